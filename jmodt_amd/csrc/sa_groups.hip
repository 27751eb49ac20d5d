// sa_groups.hip — the plan of the duplicate-aware RPN set abstraction (exact) + the listed entry of the fused SA kernels.
//
// ball_query (jmodt/ops/pointnet2/src/ball_query_gpu.cu:36-40) writes the FIRST hit into all nsample slots of a centre's
// list and then overwrites slots 1 .. cnt - 1 with the further hits, in ascending point order: a list with cnt < nsample hits
// ends in nsample - cnt copies of its first entry.  The grouped tensor of _PointnetSAModuleBase.forward
// (pointnet2_modules.py:46-61) then repeats that row, and the max-pool over the group (pointnet2_modules.py:50-52) is
// idempotent: only the first d = max(cnt, 1) rows of a group matter.  On FPS-thinned clouds most groups of the RPN levels
// hold ONE point, the centre itself (tools/rpn_dup_stats.py: 94-97 % of the rows of the headline cloud are copies).
//
//   sg_plan_kernel   one lane per (frame, centre) group: d = 1 + the last slot whose entry differs from the first (= the hit
//                    count of a ball-query list: the hits behind the first are strictly larger point indices), class
//                    q = max(qmin, ceil(log2 d)), and the group id is appended to class q's list (one atomic per workgroup
//                    and class).  The first 2^q entries of the group's list are its d distinct rows + back-fill copies.
//   consumers        sa_mlp_wide_kernel / sa_xyz_valu_kernel / sa_mlp_pm_kernel in LISTED mode: tiles of ONE class (a tile
//                    holds rows-per-tile >> q groups), pool over 2^q rows, output stored at the group's own position.
// Where a group lands in its class list is not deterministic (atomics); its value is: a row depends on (point, centre) only
// and max is order- and multiplicity-free, so the output is BIT-IDENTICAL to the dense kernels' run to run and mode to mode.
// No host decision, no host sync: the class counts live in device memory and every consumer launch is always issued.
#include "jm_common.h"

namespace jm {

int sa_mlp_wide_launch(int b, int n, int m, int c, int nsample, const float* xyz, const float* new_xyz,
                       const float* features, const int* idx, int L, const int* widths, const float* const* weights,
                       const float* const* biases, float* out, size_t obs, hipStream_t s, const int* cls_count, const int* glist);
const char* sa_wide_unsupported(long long b, int n, int m, int c, int nsample, int group_all, int L, const int* widths);
int sa_xyz_valu_launch(int b, int n, int m, int nsample, const float* xyz, const float* new_xyz, const int* idx,
                       const int* widths, const float* const* weights, const float* const* biases, float* out, size_t obs, hipStream_t s,
                       const int* cls_count, const int* glist);

// cls_count[0..7] zeroed by the caller (hipMemsetAsync in the entry); glist: (qfull + 1) x groups
struct SgPlan {                       // up to two problems over the same groups (the two scales of an MSG level): blockIdx.y
    const int* groups_dev;            // optional: the number of valid groups lives in device memory (<= groups; sa_dedupe's virtual centres)
    int ns[2], qmin[2];
    const int* idx[2];
    int* cls_count[2];
    int* glist[2];
};

// d = 1 + the LAST slot that differs from the first: for a ball-query list that is its hit count; for any other list the
// first d entries still contain every distinct entry, so the listed form is exact whatever wrote idx
template <int NS>
__device__ __forceinline__ int sg_rows_needed(const int* __restrict__ row_) {
    const int4* row = reinterpret_cast<const int4*>(row_);
    int d = 1, first = 0;
#pragma unroll
    for (int v = 0; v < NS / 4; ++v) {
        const int4 e = row[v];
        if (v == 0) first = e.x; else if (e.x != first) d = 4 * v + 1;
        if (e.y != first) d = 4 * v + 2;
        if (e.z != first) d = 4 * v + 3;
        if (e.w != first) d = 4 * v + 4;
    }
    return d;
}

__global__ void __launch_bounds__(256)
sg_plan_kernel(int groups, SgPlan pl) {
    __shared__ int wcnt[4][8], lbase[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = blockIdx.x * 256 + tid, y = blockIdx.y;
    const int NS = pl.ns[y], qmin = pl.qmin[y];
    const int* __restrict__ idx = pl.idx[y];
    int* __restrict__ cls_count = pl.cls_count[y];
    int* __restrict__ glist = pl.glist[y];
    int q = -1, rank = 0;
    const int valid = pl.groups_dev ? min(*pl.groups_dev, groups) : groups;
    if (g < valid) {
        const int* row = idx + (size_t)g * NS;
        const int d = NS == 16 ? sg_rows_needed<16>(row) : (NS == 32 ? sg_rows_needed<32>(row) : sg_rows_needed<64>(row));
        q = d <= 1 ? 0 : 32 - __clz(d - 1);                       // ceil(log2 d)
        q = max(q, qmin);
    }
    // rank within the workgroup in ascending group order (ballot prefix): a class list is then a sequence of ascending runs, so
    // neighbouring rows of a consumer's tile are neighbouring centres (coalescing loads and stores); only the ORDER OF THE RUNS
    // (one global atomic per workgroup and class) varies from run to run, which no output depends on
    for (int c = 0; c < 8; ++c) {
        const unsigned long long mask = __ballot(q == c);
        if (q == c) rank = __popcll(mask & ((1ull << lane) - 1ull));
        if (lane == 0) wcnt[wave][c] = __popcll(mask);
    }
    __syncthreads();
    if (tid < 8) {
        const int n = wcnt[0][tid] + wcnt[1][tid] + wcnt[2][tid] + wcnt[3][tid];
        if (n > 0) lbase[tid] = atomicAdd(&cls_count[tid], n);
    }
    __syncthreads();
    if (q >= 0) {
        int off = lbase[q];
        for (int w = 0; w < wave; ++w) off += wcnt[w][q];
        glist[(size_t)q * groups + off + rank] = g;
    }
}

}  // namespace jm

using namespace jm;

static int sg_nq(int nsample) { return nsample == 16 ? 5 : (nsample == 32 ? 6 : 7); }

/* ints of a class list: (log2 nsample + 1) x groups */
extern "C" size_t jm_sa_group_list_elems(int groups, int nsample) {
    if (groups < 0 || (nsample != 16 && nsample != 32 && nsample != 64)) return 0;
    return (size_t)sg_nq(nsample) * (size_t)groups;
}

static int sg_check(int groups, int nsample, const int* idx, int qmin, const int* cls_count, const int* glist) {
    JM_REQUIRE(groups >= 0 && (nsample == 16 || nsample == 32 || nsample == 64), "sa_group_plan: nsample in {16, 32, 64}");
    JM_REQUIRE(qmin >= 0 && (1 << qmin) <= nsample, "sa_group_plan: smallest class 2^%d above nsample", qmin);
    JM_REQUIRE(cls_count && (glist || groups == 0), "sa_group_plan: null plan");
    JM_REQUIRE(groups == 0 || (idx && (reinterpret_cast<uintptr_t>(idx) & 15u) == 0), "sa_group_plan: idx null or not 16-byte aligned");
    return JM_OK;
}

/* cls_count[0..7] = groups per class q (rows per group 2^q), glist[q * groups ...] = class q's group ids (b * npoint + i).
 * idx (groups, nsample) int32 as ball_query wrote it; qmin = the smallest class the consumer kernel tiles (0 = single rows). */
extern "C" int jm_sa_group_plan(int groups, int nsample, const int* idx, int qmin, int* cls_count, int* glist, jm_stream_t stream) {
    if (int rc = sg_check(groups, nsample, idx, qmin, cls_count, glist)) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (jm_zero_async(cls_count, 8 * sizeof(int), s) != hipSuccess) { set_error("sa_group_plan: memset failed"); return JM_ELAUNCH; }
    if (groups == 0) return JM_OK;
    SgPlan pl{};
    pl.ns[0] = nsample; pl.qmin[0] = qmin; pl.idx[0] = idx; pl.cls_count[0] = cls_count; pl.glist[0] = glist;
    hipLaunchKernelGGL(sg_plan_kernel, dim3((unsigned)divup(groups, 256), 1), dim3(256), 0, s, groups, pl);
    return check_launch("sa_group_plan");
}

/* the same with the number of valid groups read from device memory (groups_dev[0] <= groups; the rows behind it are never read):
 * the duplicate-compacted RCNN scales plan their virtual centres (csrc/sa_dedupe.hip), whose count is data dependent */
extern "C" int jm_sa_group_plan_dev(int groups, int nsample, const int* idx, int qmin, const int* groups_dev, int* cls_count, int* glist,
                                    jm_stream_t stream) {
    if (int rc = sg_check(groups, nsample, idx, qmin, cls_count, glist)) return rc;
    JM_REQUIRE(groups_dev, "sa_group_plan_dev: null group count");
    hipStream_t s = (hipStream_t)stream;
    if (jm_zero_async(cls_count, 8 * sizeof(int), s) != hipSuccess) { set_error("sa_group_plan: memset failed"); return JM_ELAUNCH; }
    if (groups == 0) return JM_OK;
    SgPlan pl{};
    pl.groups_dev = groups_dev;
    pl.ns[0] = nsample; pl.qmin[0] = qmin; pl.idx[0] = idx; pl.cls_count[0] = cls_count; pl.glist[0] = glist;
    hipLaunchKernelGGL(sg_plan_kernel, dim3((unsigned)divup(groups, 256), 1), dim3(256), 0, s, groups, pl);
    return check_launch("sa_group_plan_dev");
}

/* the two scales of one multi-scale level (same centres, two neighbour lists) in ONE launch: cls_count = 16 ints
 * ([0..7] scale 0, [8..15] scale 1: one memset), glist0 / glist1 as above */
extern "C" int jm_sa_group_plan_dual(int groups, int nsample0, const int* idx0, int qmin0, int nsample1, const int* idx1, int qmin1,
                                     int* cls_count, int* glist0, int* glist1, jm_stream_t stream) {
    if (int rc = sg_check(groups, nsample0, idx0, qmin0, cls_count, glist0)) return rc;
    if (int rc = sg_check(groups, nsample1, idx1, qmin1, cls_count, glist1)) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (jm_zero_async(cls_count, 16 * sizeof(int), s) != hipSuccess) { set_error("sa_group_plan: memset failed"); return JM_ELAUNCH; }
    if (groups == 0) return JM_OK;
    SgPlan pl{};
    pl.ns[0] = nsample0; pl.qmin[0] = qmin0; pl.idx[0] = idx0; pl.cls_count[0] = cls_count; pl.glist[0] = glist0;
    pl.ns[1] = nsample1; pl.qmin[1] = qmin1; pl.idx[1] = idx1; pl.cls_count[1] = cls_count + 8; pl.glist[1] = glist1;
    hipLaunchKernelGGL(sg_plan_kernel, dim3((unsigned)divup(groups, 256), 2), dim3(256), 0, s, groups, pl);
    return check_launch("sa_group_plan_dual");
}

/* which kernel takes the LISTED form of this scale (the kernel jm_sa_mlp_forward_into runs on it, so that listed == dense bit
 * for bit): 0 none, 2 sa_mlp_wide_kernel, 3 sa_xyz_valu_kernel (smallest class 2^0 = single rows for both) */
extern "C" int jm_sa_mlp_listed_supported(int b, int n, int m, int c, int nsample, int num_layers, const int* widths) {
    if (b < 0 || n < 1 || m < 1 || c < 0 || num_layers < 1 || !widths || widths[0] != 3 + c) return 0;
    const int kind = jm_sa_mlp_supported(b, n, m, c, nsample, 0, num_layers, widths);
    return kind == 2 || kind == 3 ? kind : 0;
}

extern "C" int jm_sa_mlp_listed_qmin(int kind) { return kind == 2 || kind == 3 ? 0 : -1; }

/* jm_sa_mlp_forward_into on the groups of a plan (jm_sa_group_plan of the same idx): bit-identical output */
extern "C" int jm_sa_mlp_forward_listed(int b, int n, int m, int c, int nsample, const float* xyz, const float* new_xyz,
                                        const float* features, const int* idx, int num_layers, const int* widths,
                                        const float* const* weights, const float* const* biases, const int* cls_count,
                                        const int* glist, float* out, size_t out_frame_stride, jm_stream_t stream) {
    JM_REQUIRE(b >= 0 && n >= 1 && m >= 0 && c >= 0, "sa_mlp_listed: bad sizes");
    if (b == 0 || m == 0) return JM_OK;
    JM_REQUIRE(xyz && new_xyz && idx && out && widths && weights && biases && cls_count && glist && (features || c == 0), "sa_mlp_listed: null pointer");
    JM_REQUIRE(num_layers >= 1 && widths[0] == 3 + c, "sa_mlp_listed: widths[0] = %d != 3 + C = %d", widths[0], 3 + c);
    JM_REQUIRE(out_frame_stride == 0 || out_frame_stride >= (size_t)widths[num_layers] * (size_t)m,
               "sa_mlp_listed: output frame stride below cout * npoint");
    const int kind = jm_sa_mlp_listed_supported(b, n, m, c, nsample, num_layers, widths);
    JM_REQUIRE(kind != 0, "sa_mlp_listed: no listed kernel for this shape");
    if (kind == 3)
        return sa_xyz_valu_launch(b, n, m, nsample, xyz, new_xyz, idx, widths, weights, biases, out, out_frame_stride,
                                  (hipStream_t)stream, cls_count, glist);
    return sa_mlp_wide_launch(b, n, m, c, nsample, xyz, new_xyz, features, idx, num_layers, widths, weights, biases, out,
                              out_frame_stride, (hipStream_t)stream, cls_count, glist);
}
