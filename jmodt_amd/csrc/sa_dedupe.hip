// sa_dedupe.hip — duplicate-aware set abstraction (exact): the planning / combining kernels around sa_mlp_pm_kernel.
//
// The RCNN stage's set abstraction (jmodt/detection/modeling/rcnn.py:176-200 -> pointnet2_modules.py:20-63) runs on RoI
// point sets that are full of EXACT copies: roipool3d pads a box holding cnt < 512 points cyclically (row s = row s % cnt,
// roipool3d_kernel.cu:123-160), furthest point sampling on such a set returns copies of the same point once the cnt
// distinct ones are used up, and ball_query back-fills a short neighbour list with its first hit (ball_query_gpu.cu:36-40).
// Every (centre, sample) row of the grouped tensor that repeats an earlier row of the same centre contributes nothing to
// the max-pool (max is idempotent), and a centre that is a copy of an earlier centre has the same neighbour list, hence the
// same output.  With `canon[k]` = the first point of which point k is an exact copy, the kernels here
//   plan     per RoI: centres -> representative centre (same canonical point), per representative the DISTINCT canonical
//            neighbours of its list, cut into segments of 16 rows (padded with the list's first entry) = "virtual centres"
//            of a 16-sample set-abstraction problem over the whole batch, allocated with one atomic per RoI;
//   finish   tile count = ceil(virtual centres / 8) into device memory, the last tile padded with harmless rows;
//   (sa_mlp_pm_kernel runs on the virtual centres: jm_sa_mlp_pm_forward_dyn, nothing about it changes)
//   combine  centre output = max over its representative's segments (bias and ReLU are already applied per segment and
//            commute with max).
// The result is BIT-IDENTICAL to the dense kernel's: a row's value depends on (u[point], centre) only, never on its position
// in a tile, and max over a set does not depend on multiplicity or order.  What changes is the number of rows executed.
#include "jm_common.h"

namespace jm {

constexpr int SD_SEG = 16;          // rows per virtual centre (the smallest nsample sa_mlp_pm_kernel tiles)
constexpr int SD_MAX_N = 2048;      // points per set
constexpr int SD_MAX_M = 256;       // centres per set

// canon[r][k] = k % max(cnt[r], 1): the cyclic padding of roipool3d
__global__ void sd_canon_from_cnt_kernel(int n, long long total, const int* __restrict__ cnt, int* __restrict__ canon) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int r = (int)(e / n), k = (int)(e - (long long)r * n);
    const int c = max(cnt[r], 1);
    canon[e] = k % c;
}

// canon2[r][j] = rep[r][j] of the level below: a level's points are the previous level's centres
// (nothing to compute: the caller passes `rep` itself as the next level's canon)

__global__ void __launch_bounds__(256)
sd_plan_kernel(int n, int m, int ns, int vmax, const int* __restrict__ canon, const int* __restrict__ fps_idx,
               const int* __restrict__ nb, const float* __restrict__ new_xyz, int* __restrict__ rep, int* __restrict__ seg_start,
               int* __restrict__ seg_cnt, int* __restrict__ vidx, float* __restrict__ vxyz, int* __restrict__ counter) {
    extern __shared__ int lds[];
    int* first = lds;                       // [n]      first centre using canonical point q
    int* ucnt = first + n;                  // [m]      distinct neighbours of centre i (0 for non-representatives)
    int* soff = ucnt + m;                   // [m + 1]  segment offsets within the RoI
    unsigned* bm = reinterpret_cast<unsigned*>(soff + m + 1);      // [4][n / 32 rounded up] per-wave "seen" bitmaps
    const int bw = (n + 31) >> 5;
    int* list = reinterpret_cast<int*>(bm + 4 * bw);               // [m][ns] distinct canonical neighbours per centre
    __shared__ int base_s;
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int* cn = canon + (size_t)r * n;
    for (int q = tid; q < n; q += 256) first[q] = 0x7fffffff;
    __syncthreads();
    for (int i = tid; i < m; i += 256) atomicMin(&first[cn[fps_idx[(size_t)r * m + i]]], i);
    __syncthreads();
    for (int i = tid; i < m; i += 256) {
        const int ri = first[cn[fps_idx[(size_t)r * m + i]]];
        rep[(size_t)r * m + i] = ri;
        ucnt[i] = 0;
    }
    __syncthreads();
    unsigned* mybm = bm + wave * bw;
    for (int i = wave; i < m; i += 4) {                                 // wave-uniform
        if (first[cn[fps_idx[(size_t)r * m + i]]] != i) continue;       // a copy of an earlier centre
        for (int w = lane; w < bw; w += 64) mybm[w] = 0u;
        __threadfence_block();
        int U = 0;
        for (int s0 = 0; s0 < ns; s0 += 64) {
            const int s = s0 + lane;
            bool keep = false;
            int v = 0;
            if (s < ns) {
                v = cn[nb[((size_t)r * m + i) * ns + s]];
                const unsigned bit = 1u << (v & 31);
                keep = (atomicOr(&mybm[v >> 5], bit) & bit) == 0u;      // exactly one lane per distinct value sees it unset
            }
            const unsigned long long bal = __ballot(keep);
            if (keep) list[i * ns + U + mbcnt(bal)] = v;
            U += (int)__popcll(bal);
        }
        if (lane == 0) ucnt[i] = U;
        __threadfence_block();
    }
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int i = 0; i < m; ++i) { soff[i] = run; run += (ucnt[i] + SD_SEG - 1) / SD_SEG; }
        soff[m] = run;
        base_s = atomicAdd(counter, run);
    }
    __syncthreads();
    const int base = base_s;
    for (int i = tid; i < m; i += 256) {
        const int nseg = soff[i + 1] - soff[i];
        seg_start[(size_t)r * m + i] = base + soff[i];
        seg_cnt[(size_t)r * m + i] = nseg;
    }
    // virtual centres: 16 global point indices each (padded with the list's first entry) + the centre's coordinates
    const int total_rows = soff[m] * SD_SEG;
    for (int e = tid; e < total_rows; e += 256) {
        const int sg = e / SD_SEG, t = e - sg * SD_SEG;
        int lo = 0, hi = m - 1;                                         // centre owning segment sg: last i with soff[i] <= sg and a segment
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (soff[mid] <= sg) lo = mid; else hi = mid - 1;
        }
        const int i = lo, j = sg - soff[i];
        const int pos = j * SD_SEG + t;
        const int v = list[i * ns + (pos < ucnt[i] ? pos : 0)];
        const long long ve = (long long)(base + sg);
        if (ve < vmax) vidx[ve * SD_SEG + t] = r * n + v;
        if (t < 3 && ve < vmax) vxyz[ve * 3 + t] = new_xyz[((size_t)r * m + i) * 3 + t];
    }
}

// tiles[0] = ceil(total / 8) 128-row tiles; the virtual centres that pad the last tile gather point 0 around the origin
__global__ void sd_finish_kernel(int vmax, const int* __restrict__ counter, int* __restrict__ tiles, int* __restrict__ vidx,
                                 float* __restrict__ vxyz) {
    const int total = min(counter[0], vmax);
    const int padded = min((total + 7) / 8 * 8, vmax);
    for (int e = total * SD_SEG + threadIdx.x; e < padded * SD_SEG; e += blockDim.x) vidx[e] = 0;
    for (int e = total * 3 + threadIdx.x; e < padded * 3; e += blockDim.x) vxyz[e] = 0.f;
    if (threadIdx.x == 0) { tiles[0] = padded / 8; tiles[1] = total; }
}

// out[r][col][i] = max over the segments of centre i's representative of outv[col][segment]
__global__ void __launch_bounds__(256)
sd_combine_kernel(int m, int cout, int vmax, const float* __restrict__ outv, const int* __restrict__ rep,
                  const int* __restrict__ seg_start, const int* __restrict__ seg_cnt, float* __restrict__ out) {
    const int r = blockIdx.x;
    for (int e = threadIdx.x; e < m * cout; e += blockDim.x) {
        const int col = e / m, i = e - col * m;
        const int ri = rep[(size_t)r * m + i];
        const int s0 = seg_start[(size_t)r * m + ri], nsg = seg_cnt[(size_t)r * m + ri];
        const float* q = outv + (size_t)col * vmax + s0;
        float v = q[0];
        for (int j = 1; j < nsg; ++j) v = fmaxf(v, q[j]);
        out[((size_t)r * cout + col) * m + i] = v;
    }
}

}  // namespace jm

using namespace jm;

extern "C" int jm_sa_dedupe_canon_from_cnt(int r, int n, const int* cnt, int* canon, jm_stream_t stream) {
    JM_REQUIRE(r >= 0 && n >= 1, "sa_dedupe: bad sizes");
    if (r == 0) return JM_OK;
    JM_REQUIRE(cnt && canon, "sa_dedupe: null pointer");
    const long long total = (long long)r * n;
    hipLaunchKernelGGL(sd_canon_from_cnt_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, total, cnt, canon);
    return check_launch("sa_dedupe(canon)");
}

/* capacity in virtual centres for r sets of m centres x nsample neighbours (worst case: nothing repeats), a multiple of 8 */
extern "C" long long jm_sa_dedupe_capacity(int r, int m, int nsample) {
    if (r < 0 || m < 0 || nsample < 1) return 0;
    return ((long long)r * m * ((nsample + SD_SEG - 1) / SD_SEG) + 7) / 8 * 8;
}

extern "C" int jm_sa_dedupe_plan(int r, int n, int m, int nsample, const int* canon, const int* fps_idx, const int* nb,
                                 const float* new_xyz, int* rep, int* seg_start, int* seg_cnt, int* vidx, float* vxyz,
                                 int* counters, jm_stream_t stream) {
    JM_REQUIRE(r >= 0 && n >= 1 && n <= SD_MAX_N && m >= 1 && m <= SD_MAX_M && nsample >= 1 && nsample <= 256,
               "sa_dedupe_plan: sizes out of range (n <= %d, m <= %d, nsample <= 256)", SD_MAX_N, SD_MAX_M);
    JM_REQUIRE(counters, "sa_dedupe_plan: null counters");
    hipStream_t s = (hipStream_t)stream;
    (void)jm_zero_async(counters, 4 * sizeof(int), s);
    if (r == 0) return JM_OK;
    JM_REQUIRE(canon && fps_idx && nb && new_xyz && rep && seg_start && seg_cnt && vidx && vxyz, "sa_dedupe_plan: null pointer");
    const long long vmax = jm_sa_dedupe_capacity(r, m, nsample);
    JM_REQUIRE(vmax * SD_SEG < (1LL << 31) && (long long)r * n < (1LL << 31), "sa_dedupe_plan: too many rows");
    const size_t lds = ((size_t)n + m + (m + 1) + 4 * ((n + 31) / 32) + (size_t)m * nsample) * sizeof(int);
    JM_REQUIRE(lds <= 150 * 1024, "sa_dedupe_plan: set too large for the LDS");
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)sd_plan_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(sd_plan_kernel, dim3((unsigned)r), dim3(256), lds, s, n, m, nsample, (int)vmax, canon, fps_idx, nb, new_xyz, rep,
                       seg_start, seg_cnt, vidx, vxyz, counters);
    hipLaunchKernelGGL(sd_finish_kernel, dim3(1), dim3(256), 0, s, (int)vmax, counters, counters + 1, vidx, vxyz);
    return check_launch("sa_dedupe_plan");
}

extern "C" int jm_sa_dedupe_combine(int r, int m, int cout, long long vmax, const float* out_virtual, const int* rep,
                                    const int* seg_start, const int* seg_cnt, float* out, jm_stream_t stream) {
    JM_REQUIRE(r >= 0 && m >= 1 && cout >= 1 && vmax >= 0 && vmax < (1LL << 31), "sa_dedupe_combine: bad sizes");
    if (r == 0) return JM_OK;
    JM_REQUIRE(out_virtual && rep && seg_start && seg_cnt && out, "sa_dedupe_combine: null pointer");
    hipLaunchKernelGGL(sd_combine_kernel, dim3((unsigned)r), dim3(256), 0, (hipStream_t)stream, m, cout, (int)vmax, out_virtual, rep,
                       seg_start, seg_cnt, out);
    return check_launch("sa_dedupe_combine");
}
