// iou3d.hip — rotated / axis-aligned BEV overlap, IoU and NMS for gfx950.
//
// Replaces boxes_overlap_kernel, boxes_iou_bev_kernel, nms_kernel, nms_normal_kernel
// (jmodt/ops/iou3d/src/iou3d_kernel.cu:223-348) and the HOST greedy reduce of nms_gpu /
// nms_normal_gpu (iou3d.cpp:73-166).
//
// Design:
//  * Per-box work is hoisted out of the pair loop: the rotated corners, centre and sin/cos of a
//    box are computed once per workgroup tile and kept in LDS / registers (the reference
//    recomputes cos/sin and the 4 rotated corners of BOTH boxes for every pair).
//  * NMS never leaves the device: the 64x64-tile suppression mask is built for the upper
//    triangle only (the reference launches the full square and discards half), and the greedy
//    reduce runs as a single workgroup — wave 0 resolves the 64 boxes of a row block with a
//    scalar loop over the diagonal tile (readlane, no memory), then all threads OR the kept
//    rows into the removal mask held in LDS.  The reference cudaMalloc's, memcpy's 5 MB to the
//    host, loops serially on the CPU and cudaFree's, twice per frame.
//  * All arithmetic is the reference's float expression sequence with contraction off;
//    sin/cos/atan2 through include/jm_detmath.h so the keep set is bit-identical to the oracle.
#include "../../include/jm_detmath.h"
#include "jm_common.h"

namespace jm {

struct Pt { float x, y; };

struct RBox {           // per-box precomputation
    float x1, y1, x2, y2;
    float cs, sn;       // cos / sin of the heading
    Pt c;               // centre
    Pt corner[4];       // rotated corners
    float area;
};

__device__ __forceinline__ float cross3(Pt p1, Pt p2, Pt p0) {
    return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}

__device__ __forceinline__ Pt rotate_pt(Pt center, float c, float s, Pt p) {
    Pt r;
    r.x = (p.x - center.x) * c + (p.y - center.y) * s + center.x;
    r.y = -(p.x - center.x) * s + (p.y - center.y) * c + center.y;
    return r;
}

__device__ __forceinline__ RBox make_rbox(const float* b) {
    RBox r;
    r.x1 = b[0]; r.y1 = b[1]; r.x2 = b[2]; r.y2 = b[3];
    jm_sincosf(b[4], &r.sn, &r.cs);
    r.c.x = (r.x1 + r.x2) / 2; r.c.y = (r.y1 + r.y2) / 2;
    const Pt raw[4] = {{r.x1, r.y1}, {r.x2, r.y1}, {r.x2, r.y2}, {r.x1, r.y2}};
#pragma unroll
    for (int k = 0; k < 4; ++k) r.corner[k] = rotate_pt(r.c, r.cs, r.sn, raw[k]);
    r.area = (r.x2 - r.x1) * (r.y2 - r.y1);
    return r;
}

// check_in_box2d (iou3d_kernel.cu:50-65): rotate p by -angle (cos(-a)=cos a, sin(-a)=-sin a)
__device__ __forceinline__ int in_box2d(const RBox& b, Pt p) {
    const float MARGIN = 1e-5f;
    const float angle_cos = b.cs, angle_sin = -b.sn;
    const float rot_x = (p.x - b.c.x) * angle_cos + (p.y - b.c.y) * angle_sin + b.c.x;
    const float rot_y = -(p.x - b.c.x) * angle_sin + (p.y - b.c.y) * angle_cos + b.c.y;
    return (rot_x > b.x1 - MARGIN && rot_x < b.x2 + MARGIN && rot_y > b.y1 - MARGIN && rot_y < b.y2 + MARGIN);
}

// intersection (iou3d_kernel.cu:67-96)
__device__ __forceinline__ int seg_intersection(Pt p1, Pt p0, Pt q1, Pt q0, Pt& ans) {
    const int rect = fminf(p0.x, p1.x) <= fmaxf(q0.x, q1.x) && fminf(q0.x, q1.x) <= fmaxf(p0.x, p1.x) &&
                     fminf(p0.y, p1.y) <= fmaxf(q0.y, q1.y) && fminf(q0.y, q1.y) <= fmaxf(p0.y, p1.y);
    if (!rect) return 0;
    const float s1 = cross3(q0, p1, p0);
    const float s2 = cross3(p1, q1, p0);
    const float s3 = cross3(p0, q1, q0);
    const float s4 = cross3(q1, p1, q0);
    if (!(s1 * s2 > 0 && s3 * s4 > 0)) return 0;
    const float s5 = cross3(q1, p1, p0);
    if (fabs((double)(s5 - s1)) > 1e-8) {
        ans.x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
        ans.y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
    } else {
        const float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
        const float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
        const float D = a0 * b1 - a1 * b0;
        ans.x = (b0 * c1 - b1 * c0) / D;
        ans.y = (a1 * c0 - a0 * c1) / D;
    }
    return 1;
}

// box_overlap (iou3d_kernel.cu:108-212) on precomputed boxes
__device__ float rbox_overlap(const RBox& A, const RBox& B) {
    Pt cp[24];
    float ang[24];
    Pt center = {0.f, 0.f};
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            Pt ans;
            if (seg_intersection(A.corner[(i + 1) & 3], A.corner[i], B.corner[(j + 1) & 3], B.corner[j], ans)) {
                cp[cnt] = ans;
                center.x = center.x + ans.x;
                center.y = center.y + ans.y;
                ++cnt;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (in_box2d(A, B.corner[k])) {
            center.x = center.x + B.corner[k].x; center.y = center.y + B.corner[k].y;
            cp[cnt++] = B.corner[k];
        }
        if (in_box2d(B, A.corner[k])) {
            center.x = center.x + A.corner[k].x; center.y = center.y + A.corner[k].y;
            cp[cnt++] = A.corner[k];
        }
    }
    if (cnt == 0) return 0.f;
    center.x /= cnt;
    center.y /= cnt;
    for (int i = 0; i < cnt; ++i) ang[i] = jm_atan2f(cp[i].y - center.y, cp[i].x - center.x);
    for (int j = 0; j < cnt - 1; ++j)
        for (int i = 0; i < cnt - j - 1; ++i)
            if (ang[i] > ang[i + 1]) {
                const Pt t = cp[i]; cp[i] = cp[i + 1]; cp[i + 1] = t;
                const float ta = ang[i]; ang[i] = ang[i + 1]; ang[i + 1] = ta;
            }
    float area = 0.f;
    for (int k = 0; k < cnt - 1; ++k) {
        const float ux = cp[k].x - cp[0].x, uy = cp[k].y - cp[0].y;
        const float vx = cp[k + 1].x - cp[0].x, vy = cp[k + 1].y - cp[0].y;
        area += ux * vy - uy * vx;
    }
    return (float)(fabs((double)area) / 2.0);
}

__device__ __forceinline__ float rbox_iou(const RBox& A, const RBox& B) {
    const float s = rbox_overlap(A, B);
    return s / fmaxf(A.area + B.area - s, (float)1e-8);
}

// iou_normal (iou3d_kernel.cu:295-303)
__device__ __forceinline__ float iou_normal(const float* a, const float* b) {
    const float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]);
    const float top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
    const float width = fmaxf(right - left, 0.f), height = fmaxf(bottom - top, 0.f);
    const float interS = width * height;
    const float Sa = (a[2] - a[0]) * (a[3] - a[1]);
    const float Sb = (b[2] - b[0]) * (b[3] - b[1]);
    return interS / fmaxf(Sa + Sb - interS, (float)1e-8);
}

// ------------------------------------------------------------------ pairwise overlap / IoU
// block (16 x 16): threadIdx.y -> a, threadIdx.x -> b, per-box precompute shared through LDS
template <bool IOU>
__global__ void __launch_bounds__(256)
boxes_pair_kernel(int num_a, const float* __restrict__ boxes_a, int num_b, const float* __restrict__ boxes_b,
                  float* __restrict__ ans) {
    __shared__ RBox sa[16], sb[16];
    const int a0 = blockIdx.y * 16, b0 = blockIdx.x * 16;
    const int t = threadIdx.y * 16 + threadIdx.x;
    if (t < 16) { if (a0 + t < num_a) sa[t] = make_rbox(boxes_a + (size_t)(a0 + t) * 5); }
    else if (t < 32) { if (b0 + t - 16 < num_b) sb[t - 16] = make_rbox(boxes_b + (size_t)(b0 + t - 16) * 5); }
    __syncthreads();
    const int ai = a0 + threadIdx.y, bi = b0 + threadIdx.x;
    if (ai >= num_a || bi >= num_b) return;
    ans[(size_t)ai * num_b + bi] = IOU ? rbox_iou(sa[threadIdx.y], sb[threadIdx.x]) : rbox_overlap(sa[threadIdx.y], sb[threadIdx.x]);
}

// ------------------------------------------------------------------ tracker association cost
// cost[i,j] = link[i,j] * w_app + iou3d(pred_i, det_j) * w_iou + dist(pred_i, det_j) * w_dis
// (jmodt/tracking/data_association.py:10-28,42-45,117-119): rotated BEV overlap x height overlap over
// the volume union, and 1 - centre distance / farthest of the 64 corner-pair distances.  The
// reference builds this from ~25 torch kernels and two (M,N,8,8,3) repeat() temporaries, then
// copies it to the host; here it is one launch, per-box corners computed once per tile.
struct Box3 {
    RBox bev;
    float cx, cy, cz, h, vol;
    float corner[8][3];
};

__device__ __forceinline__ Box3 make_box3(const float* b) {
    Box3 r;
    const float x = b[0], y = b[1], z = b[2], h = b[3], w = b[4], l = b[5];
    // boxes3d_to_bev_torch (kitti_utils.py:136-149)
    const float bev[5] = {x - l / 2, z - w / 2, x + l / 2, z + w / 2, b[6]};
    r.bev = make_rbox(bev);
    r.cx = x; r.cy = y; r.cz = z; r.h = h; r.vol = h * w * l;
    const float xs[8] = {l / 2, l / 2, -l / 2, -l / 2, l / 2, l / 2, -l / 2, -l / 2};
    const float ys[8] = {0.f, 0.f, 0.f, 0.f, -h, -h, -h, -h};
    const float zs[8] = {w / 2, -w / 2, -w / 2, w / 2, w / 2, -w / 2, -w / 2, w / 2};
#pragma unroll
    for (int i = 0; i < 8; ++i) {   // boxes3d_to_corners3d_torch (kitti_utils.py:107-133)
        r.corner[i][0] = r.bev.cs * xs[i] + r.bev.sn * zs[i] + x;
        r.corner[i][1] = ys[i] + y;
        r.corner[i][2] = -r.bev.sn * xs[i] + r.bev.cs * zs[i] + z;
    }
    return r;
}

__global__ void __launch_bounds__(256)
association_cost_kernel(int np, const float* __restrict__ pred, int nd, const float* __restrict__ det,
                        const float* __restrict__ link, float w_app, float w_iou, float w_dis,
                        float* __restrict__ cost, float* __restrict__ iou_out, float* __restrict__ dist_out) {
    __shared__ Box3 sa[16], sb[16];
    const int a0 = blockIdx.y * 16, b0 = blockIdx.x * 16;
    const int t = threadIdx.y * 16 + threadIdx.x;
    if (t < 16) { if (a0 + t < np) sa[t] = make_box3(pred + (size_t)(a0 + t) * 7); }
    else if (t < 32) { if (b0 + t - 16 < nd) sb[t - 16] = make_box3(det + (size_t)(b0 + t - 16) * 7); }
    __syncthreads();
    const int ai = a0 + threadIdx.y, bi = b0 + threadIdx.x;
    if (ai >= np || bi >= nd) return;
    const Box3& A = sa[threadIdx.y];
    const Box3& B = sb[threadIdx.x];
    // iou3d_utils.py:36-52 (y is the box BOTTOM, y - h the top)
    const float ov_bev = rbox_overlap(A.bev, B.bev);
    const float hmin = fmaxf(A.cy - A.h, B.cy - B.h), hmax = fminf(A.cy, B.cy);
    const float ov3 = ov_bev * fmaxf(hmax - hmin, 0.f);
    const float iou = ov3 / fmaxf(A.vol + B.vol - ov3, 1e-7f);
    const float dx = A.cx - B.cx, dy = A.cy - B.cy, dz = A.cz - B.cz;
    const float centre = sqrtf(dx * dx + dy * dy + dz * dz);
    float far2 = 0.f;
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float ex = A.corner[p][0] - B.corner[q][0], ey = A.corner[p][1] - B.corner[q][1],
                        ez = A.corner[p][2] - B.corner[q][2];
            far2 = fmaxf(far2, ex * ex + ey * ey + ez * ez);
        }
    const float dist = 1.f - centre / sqrtf(far2);
    const size_t o = (size_t)ai * nd + bi;
    if (iou_out) iou_out[o] = iou;
    if (dist_out) dist_out[o] = dist;
    if (cost) cost[o] = (link ? link[o] * w_app : 0.f) + iou * w_iou + dist * w_dis;
}

// ------------------------------------------------------------------ NMS mask (upper triangle)
// Batched over blockIdx.y: problem p has counts[p] (<= nmax) score-sorted boxes at boxes + p*nmax*5 and a
// mask slab of nmax rows x stride_cb words.  counts == nullptr: one problem of exactly nmax boxes.
template <bool NORMAL>
__global__ void __launch_bounds__(64)
nms_mask_kernel(int nmax, const int* __restrict__ counts, int stride_cb, float thresh,
                const float* __restrict__ boxes_all, unsigned long long* __restrict__ mask_all) {
    const int prob = blockIdx.y;
    const int n = counts ? min(counts[prob], nmax) : nmax;
    const float* boxes = boxes_all + (size_t)prob * nmax * 5;
    unsigned long long* mask = mask_all + (size_t)prob * nmax * stride_cb;
    const int col_blocks = (nmax + 63) / 64;   // tile enumeration over the padded size
    // linear block id -> (row_start <= col_start) pair of the upper triangle
    int rb = 0, cb = 0;
    {
        // row r owns (col_blocks - r) tiles; find r by solving the triangular number, then fix up
        long long id = blockIdx.x;
        const double cbd = (double)col_blocks;
        int r = (int)floor((2.0 * cbd + 1.0 - sqrt((2.0 * cbd + 1.0) * (2.0 * cbd + 1.0) - 8.0 * (double)id)) / 2.0);
        if (r < 0) r = 0;
        if (r >= col_blocks) r = col_blocks - 1;
        auto start_of = [&](int rr) { return (long long)rr * col_blocks - (long long)rr * (rr - 1) / 2; };
        while (r > 0 && start_of(r) > id) --r;
        while (r + 1 < col_blocks && start_of(r + 1) <= id) ++r;
        rb = r; cb = r + (int)(id - start_of(r));
    }
    if (rb * 64 >= n || cb * 64 >= n) return;   // tile lies in the padding of this problem
    const int tx = threadIdx.x;
    const int row_size = min(n - rb * 64, 64), col_size = min(n - cb * 64, 64);
    __shared__ RBox scol[NORMAL ? 1 : 64];
    __shared__ float sraw[64 * 5];
    if (tx < col_size) {
        const float* src = boxes + (size_t)(cb * 64 + tx) * 5;
        if (NORMAL) {
#pragma unroll
            for (int q = 0; q < 5; ++q) sraw[tx * 5 + q] = src[q];
        } else {
            scol[tx] = make_rbox(src);
        }
    }
    __syncthreads();
    if (tx < row_size) {
        const int cur = rb * 64 + tx;
        const float* cur_box = boxes + (size_t)cur * 5;
        unsigned long long t = 0;
        const int start = (rb == cb) ? tx + 1 : 0;
        if (NORMAL) {
            const float cb5[5] = {cur_box[0], cur_box[1], cur_box[2], cur_box[3], cur_box[4]};
            for (int i = start; i < col_size; ++i)
                if (iou_normal(cb5, sraw + i * 5) > thresh) t |= 1ULL << i;
        } else {
            const RBox me = make_rbox(cur_box);
            for (int i = start; i < col_size; ++i)
                if (rbox_iou(me, scol[i]) > thresh) t |= 1ULL << i;
        }
        mask[(size_t)cur * stride_cb + cb] = t;
    }
}

// The same mask at LANE granularity (round 4), for the small problems whose time is latency: the detections' rotated NMS is 8
// frames x <= 128 boxes = 24 tiles, and the tile kernel above gives each of its 64 threads a row to walk through 64
// polygon clips one after the other (0.32 ms for 131 k pairs).  Here a WAVE owns one mask word: (row, column block), lane l
// evaluates the pair (row, 64 cb + l) — the same make_rbox / rbox_iou (iou_normal) call with the row box first, so the
// predicate is bit for bit the tile kernel's — and the word is the wave's ballot.  Same words written (upper triangle incl.
// the diagonal block), n * ceil(n / 64) waves per problem; the tile kernel stays for large n, where preparing a column
// block's 64 boxes once per tile instead of once per row is the cheaper way (launch_nms_mask picks).
template <bool NORMAL>
__global__ void __launch_bounds__(256)
nms_mask_lane_kernel(int nmax, const int* __restrict__ counts, int stride_cb, float thresh,
                     const float* __restrict__ boxes_all, unsigned long long* __restrict__ mask_all) {
    const int prob = blockIdx.y;
    const int n = counts ? min(counts[prob], nmax) : nmax;
    const float* boxes = boxes_all + (size_t)prob * nmax * 5;
    unsigned long long* mask = mask_all + (size_t)prob * nmax * stride_cb;
    const int col_blocks = (nmax + 63) / 64;
    const int lane = threadIdx.x & 63;
    const int unit = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    const int row = unit / col_blocks, cb = unit - row * col_blocks;
    if (row >= n || cb * 64 >= n || cb < (row >> 6)) return;       // padding, or below the diagonal block (never read)
    const int col = cb * 64 + lane;
    const bool valid = col < n && col > row;
    const float* rb5 = boxes + (size_t)row * 5;
    const float* cb5 = boxes + (size_t)min(col, n - 1) * 5;        // clamped: every lane evaluates a real pair
    bool hit;
    if (NORMAL) {
        const float a[5] = {rb5[0], rb5[1], rb5[2], rb5[3], rb5[4]};
        const float b[5] = {cb5[0], cb5[1], cb5[2], cb5[3], cb5[4]};
        hit = iou_normal(a, b) > thresh;
    } else {
        const RBox me = make_rbox(rb5);
        const RBox other = make_rbox(cb5);
        hit = rbox_iou(me, other) > thresh;
    }
    const unsigned long long word = __ballot(valid && hit);
    if (lane == 0) mask[(size_t)row * stride_cb + cb] = word;
}

// ------------------------------------------------------------------ NMS greedy reduce (device)
// Single workgroup of 1024 threads.  remv (col_blocks x u64) lives in LDS.  Semantics of
// iou3d.cpp:98-114: ascending i, keep i if its bit is not set in remv, then
// remv |= mask_row(i) for column blocks >= i/64.
//
// The row blocks are inherently sequential (whether box i survives depends on every kept box
// before it), so the kernel is built around the latency of one row-block step:
//  * wave 0 resolves the 64 boxes of the block with a scalar loop over the *surviving
//    candidates* only (ctz + readlane of the diagonal tile; no memory access);
//  * the mask words the OR phase will need for row block rb+1 (64 rows x up to 128 later column
//    blocks, 8 words per thread, coalesced along the row) and wave 0's next diagonal tile are
//    PREFETCHED into registers during step rb, before it is known which rows survive; the
//    barriers are raw s_barrier + lgkmcnt(0) so those global loads stay in flight across them
//    (a __syncthreads() would drain vmcnt).
//  * selected words are OR-ed in registers and merged with LDS atomics (ds_or_b64).
constexpr int NMS_LANE_MAX = 1024;   // up to here the pair mask is built one pair per lane (nms_mask_lane_kernel)
constexpr int NMS_RT = 512;   // 4 row groups x 128 columns; 256-VGPR budget holds the 4-deep prefetch ring

__global__ void __launch_bounds__(NMS_RT)
nms_reduce_kernel(int nmax, const int* __restrict__ counts, int stride_cb,
                  const unsigned long long* __restrict__ mask_all, long long* __restrict__ keep_all,
                  int* __restrict__ num_keep_all) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long remv[];  // [col_blocks] + 1 (kept bits)
    const int prob = blockIdx.x;
    const int n = counts ? min(counts[prob], nmax) : nmax;
    const unsigned long long* mask = mask_all + (size_t)prob * nmax * stride_cb;
    long long* keep = keep_all + (size_t)prob * nmax;
    int* num_keep = num_keep_all + prob;
    if (n <= 0) { if (threadIdx.x == 0) *num_keep = 0; return; }
    const int cb = (n + 63) / 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int c = tid & 127, g = tid >> 7;   // OR phase: column offset 0..127, row group 0..3 (rows g + 4q)
    for (int j = tid; j <= cb; j += NMS_RT) remv[j] = 0ULL;

    // ring of 4 register sets: the words of row blocks rb .. rb+3 are in flight / resident, so a
    // load has ~3 row-block steps (resolve + 2 barriers each) to arrive from L2/HBM
    unsigned long long W[4][16];
    unsigned long long D[4] = {0ULL, 0ULL, 0ULL, 0ULL};  // wave 0: diagonal tile row of this lane
    // Loads are UNCONDITIONAL on clamped (always in-bounds) addresses: a `cond ? load : 0` makes
    // hipcc branch around every load and drain vmcnt(0) at each join.  Out-of-range words are
    // never consumed: rows >= n cannot be kept (rowmask), columns >= cb are skipped at the OR.
    auto prefetch = [&](int rb, unsigned long long (&w)[16], unsigned long long& d) {
        const int rbc = min(rb, cb - 1);
        const int col = min(rbc + 1 + c, cb - 1);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int row = min(rbc * 64 + g + 4 * q, n - 1);
            w[q] = mask[(size_t)row * stride_cb + col];
        }
        d = mask[(size_t)min(rbc * 64 + lane, n - 1) * stride_cb + rbc];   // consumed by wave 0 only
    };
    int count = 0;  // uniform
    auto step = [&](int rb, unsigned long long (&w)[16], unsigned long long& d) {
        const int rows = min(64, n - rb * 64);
        if (tid < 64) {  // wave 0: resolve this row block
            const unsigned long long r = remv[rb];
            const unsigned long long rowmask = rows == 64 ? ~0ULL : ((1ULL << rows) - 1ULL);
            unsigned long long cand = ~r & rowmask, kept = 0ULL;
            // Greedy resolution of the block in PARALLEL ROUNDS instead of one candidate at a time:
            // a candidate that no other current candidate suppresses can never be removed (everything
            // that could remove it is itself a candidate with a lower index), so all such boxes are
            // kept at once and what they suppress is dropped.  The lowest candidate is always safe,
            // so every round makes progress; rounds = depth of the suppression chains (1-3 in
            // practice) instead of one ~80-cycle scalar step per kept box.
            const unsigned long long me = 1ULL << lane;
            while (cand) {
                const unsigned long long mine = (cand & me) ? (d & cand) : 0ULL;   // candidates this lane would suppress
                const unsigned long long hit = wave_or_u64(mine);
                const unsigned long long safe = cand & ~hit;
                kept |= safe;
                const unsigned long long drop = wave_or_u64((safe & me) ? d : 0ULL);
                cand &= ~(safe | drop);
            }
            if ((kept >> lane) & 1ULL) keep[count + __popcll(kept & ((1ULL << lane) - 1ULL))] = rb * 64 + lane;
            if (lane == 0) remv[cb] = kept;
        }
        lds_barrier();
        const unsigned long long kept = remv[cb];
        count += __popcll(kept);
        // OR the kept rows into the later column blocks: first 128 columns from the prefetched words
        {
            unsigned long long acc = 0ULL;
#pragma unroll
            for (int q = 0; q < 16; ++q)
                if ((kept >> (g + 4 * q)) & 1ULL) acc |= w[q];
            const int col = rb + 1 + c;
            if (acc != 0ULL && col < cb) atomicOr(&remv[col], acc);
        }
        // columns beyond the prefetch window (n > ~8200 boxes): plain loads
        for (int col = rb + 1 + 128 + c; col < cb; col += 128) {
            unsigned long long acc = 0ULL;
            for (int q = 0; q < 16; ++q) {
                const int i = g + 4 * q;
                if (((kept >> i) & 1ULL) && rb * 64 + i < n) acc |= mask[(size_t)(rb * 64 + i) * stride_cb + col];
            }
            if (acc != 0ULL) atomicOr(&remv[col], acc);
        }
        prefetch(rb + 4, w, d);   // refill this ring slot; stays in flight across the barriers
        lds_barrier();
    };
    prefetch(0, W[0], D[0]);
    prefetch(1, W[1], D[1]);
    prefetch(2, W[2], D[2]);
    prefetch(3, W[3], D[3]);
    __syncthreads();
    for (int rb = 0; rb < cb; rb += 4) {
        step(rb, W[0], D[0]);
        if (rb + 1 < cb) step(rb + 1, W[1], D[1]);
        if (rb + 2 < cb) step(rb + 2, W[2], D[2]);
        if (rb + 3 < cb) step(rb + 3, W[3], D[3]);
    }
    if (tid == 0) *num_keep = count;
}

// ------------------------------------------------------------------ NMS, first K survivors only (axis-aligned IoU)
// The RPN keeps at most post_nms_top_n boxes per depth band (proposal_layer.py:103-117: `keep_idx[:post_nms]`), i.e. only the
// FIRST K entries of the keep list are ever read — and the greedy rule (iou3d.cpp:98-114: box i survives iff no KEPT box
// before it overlaps it by more than the threshold) needs, for box i, its IoU with the kept boxes only.  So instead of the
// n^2/2 pair mask (19.8 M evaluations for 6300 boxes) + reduce, one workgroup per problem walks the score order in chunks:
//   phase A (all waves)      every candidate of the chunk against the boxes kept before the chunk (LDS broadcast reads);
//   phase B (wave by wave)   the candidates that are still alive against the boxes kept earlier IN this chunk, then the wave
//                            resolves its own 64 in order: lowest alive lane is kept, published to LDS, the rest test
//                            against it (one ~300-cycle step per KEPT box, <= K of them in total);
// and stops at K survivors.  Evaluations: <= n * K (560 k at n = 6300, K = 89), typically a few thousand.  The IoU is
// nms_mask_kernel's own expression with the same operand order (kept box first), so the keep list is the first K entries of
// the mask + reduce result bit for bit (tests/test_gpu_parity.py::test_nms_first_k_*).
constexpr int NFK_W = 8, NFK_T = NFK_W * 64, NFK_CAP = 2048;

__global__ void __launch_bounds__(NFK_T)
nms_first_k_kernel(int nmax, const int* __restrict__ counts, float thresh, const float* __restrict__ boxes_all, int group,
                   int cap0, int cap1, long long* __restrict__ keep_all, int* __restrict__ num_keep_all,
                   unsigned long long* __restrict__ evals_all) {
    __shared__ float4 kb[NFK_CAP];        // x1, y1, x2, y2 of the kept boxes (iou_normal ignores ry)
    __shared__ int nk_hist[NFK_W + 1];    // kept count after wave w's turn (slot w + 1); one slot per turn: no reuse race
    const int prob = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = counts ? min(counts[prob], nmax) : nmax;
    const int cap = min(min((prob % group) ? cap1 : cap0, n), NFK_CAP);
    const float* boxes = boxes_all + (size_t)prob * nmax * 5;
    long long* keep = keep_all + (size_t)prob * nmax;
    unsigned int tests = 0;
    int nk = 0;   // uniform
    if (n > 0 && cap > 0) {
        for (int base = 0; base < n; base += NFK_T) {
            const int c = base + tid;
            const float* src = boxes + (size_t)min(c, n - 1) * 5;
            const float me[4] = {src[0], src[1], src[2], src[3]};
            bool alive = c < n;
            const int K0 = nk;
            // candidate `me` against kept boxes [k0, k1): four per step with independent IoUs (a single wave has nothing else to
            // hide the dependent-issue latency with), one wave-wide "anyone left?" check per step
            auto pretest = [&](int k0, int k1) {
                int k = k0;
                for (; k + 4 <= k1; k += 4) {
                    if (__ballot(alive) == 0ULL) return;
                    bool hit = false;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float4 q4 = kb[k + u];
                        const float q[4] = {q4.x, q4.y, q4.z, q4.w};
                        hit |= iou_normal(q, me) > thresh;
                    }
                    tests += alive ? 4u : 0u;
                    alive = alive && !hit;
                }
                for (; k < k1; ++k) {
                    const float4 q4 = kb[k];
                    const float q[4] = {q4.x, q4.y, q4.z, q4.w};
                    tests += alive ? 1u : 0u;
                    alive = alive && !(iou_normal(q, me) > thresh);
                }
            };
            pretest(0, K0);                                                  // phase A
            for (int w = 0; w < NFK_W; ++w) {                                // phase B
                if (wave == w) {
                    int cur = w == 0 ? K0 : nk_hist[w];
                    pretest(K0, cur);
                    unsigned long long am = __ballot(alive);
                    while (am != 0ULL && cur < cap) {
                        const int j = (int)__ffsll((long long)am) - 1;
                        am &= am - 1ULL;
                        if (lane == j) {
                            kb[cur] = make_float4(me[0], me[1], me[2], me[3]);
                            keep[cur] = c;
                        }
                        // box j to every lane through the scalar registers (v_readlane with a uniform index): no LDS round trip
                        const float q[4] = {__int_as_float(__builtin_amdgcn_readlane(__float_as_int(me[0]), j)),
                                            __int_as_float(__builtin_amdgcn_readlane(__float_as_int(me[1]), j)),
                                            __int_as_float(__builtin_amdgcn_readlane(__float_as_int(me[2]), j)),
                                            __int_as_float(__builtin_amdgcn_readlane(__float_as_int(me[3]), j))};
                        const bool mine = (am >> lane) & 1ULL;
                        tests += mine ? 1u : 0u;
                        am &= ~__ballot(mine && iou_normal(q, me) > thresh);
                        ++cur;
                    }
                    if (lane == 0) nk_hist[w + 1] = cur;
                }
                __syncthreads();
                nk = nk_hist[w + 1];
                if (nk >= cap) break;                                        // uniform
            }
            if (nk >= cap) break;
        }
    }
    if (tid == 0) num_keep_all[prob] = nk;
    if (evals_all) {
        const unsigned long long tot = (unsigned long long)wave_sum_u32(tests);
        if (lane == 0 && tot) atomicAdd(evals_all + prob, tot);
    }
}

int launch_nms_first_k(int nprob, int nmax, const int* counts, const float* boxes, float thresh, int group, int cap0, int cap1,
                       int64_t* keep, int* num_keep, unsigned long long* evals, hipStream_t s) {
    JM_REQUIRE(group >= 1 && cap0 >= 0 && cap1 >= 0 && cap0 <= NFK_CAP && cap1 <= NFK_CAP,
               "nms_first_k: at most %d survivors per problem", NFK_CAP);
    if (evals) (void)jm_zero_async(evals, sizeof(unsigned long long) * nprob, s);
    hipLaunchKernelGGL(nms_first_k_kernel, dim3(nprob), dim3(NFK_T), 0, s, nmax, counts, thresh, boxes, group, cap0, cap1,
                       (long long*)keep, num_keep, evals);
    return check_launch("nms_first_k");
}
int nms_first_k_capacity() { return NFK_CAP; }

}  // namespace jm

using namespace jm;

extern "C" int jm_boxes_overlap_bev(int num_a, const float* boxes_a, int num_b, const float* boxes_b,
                                    float* ans_overlap, jm_stream_t stream) {
    JM_REQUIRE(num_a >= 0 && num_b >= 0, "boxes_overlap_bev: bad sizes");
    if (num_a == 0 || num_b == 0) return JM_OK;
    JM_REQUIRE(boxes_a && boxes_b && ans_overlap, "boxes_overlap_bev: null pointer");
    hipLaunchKernelGGL(boxes_pair_kernel<false>, dim3(divup(num_b, 16), divup(num_a, 16)), dim3(16, 16), 0,
                       (hipStream_t)stream, num_a, boxes_a, num_b, boxes_b, ans_overlap);
    return check_launch("boxes_overlap_bev");
}

extern "C" int jm_boxes_iou_bev(int num_a, const float* boxes_a, int num_b, const float* boxes_b, float* ans_iou,
                                jm_stream_t stream) {
    JM_REQUIRE(num_a >= 0 && num_b >= 0, "boxes_iou_bev: bad sizes");
    if (num_a == 0 || num_b == 0) return JM_OK;
    JM_REQUIRE(boxes_a && boxes_b && ans_iou, "boxes_iou_bev: null pointer");
    hipLaunchKernelGGL(boxes_pair_kernel<true>, dim3(divup(num_b, 16), divup(num_a, 16)), dim3(16, 16), 0,
                       (hipStream_t)stream, num_a, boxes_a, num_b, boxes_b, ans_iou);
    return check_launch("boxes_iou_bev");
}

extern "C" int jm_association_cost(int num_pred, const float* pred_boxes, int num_det, const float* det_boxes,
                                   const float* link_scores, float w_app, float w_iou, float w_dis, float* cost,
                                   float* iou3d_out, float* dist_out, jm_stream_t stream) {
    JM_REQUIRE(num_pred >= 0 && num_det >= 0, "association_cost: bad sizes");
    if (num_pred == 0 || num_det == 0) return JM_OK;
    JM_REQUIRE(pred_boxes && det_boxes && (cost || iou3d_out || dist_out), "association_cost: null pointer");
    hipLaunchKernelGGL(association_cost_kernel, dim3(divup(num_det, 16), divup(num_pred, 16)), dim3(16, 16), 0,
                       (hipStream_t)stream, num_pred, pred_boxes, num_det, det_boxes, link_scores, w_app, w_iou, w_dis,
                       cost, iou3d_out, dist_out);
    return check_launch("association_cost");
}

// ------------------------------------------------------------------ batched 3D IoU (RoI sampling)
// boxes_iou3d_gpu (iou3d_utils.py:25-54) for every frame of a batch in one launch: the training-time RoI
// sampler calls it once per frame in a Python loop (proposal_target_layer.py:137-151, :288), each call being
// two BEV conversions, one overlap kernel and ~12 torch element-wise kernels.  blockIdx.z = frame; counts_b[f]
// (device, may be NULL) is the number of valid boxes of frame f in boxes_b (ground-truth lists are zero-padded,
// proposal_target_layer.py:141-145); columns beyond it are written as 0.
namespace jm {
__global__ void __launch_bounds__(256)
iou3d_batched_kernel(int na, const float* __restrict__ boxes_a, int nb, const float* __restrict__ boxes_b,
                     const int* __restrict__ counts_b, float* __restrict__ iou_out) {
    __shared__ Box3 sa[16], sb[16];
    const int f = blockIdx.z;
    const float* A0 = boxes_a + (size_t)f * na * 7;
    const float* B0 = boxes_b + (size_t)f * nb * 7;
    const int nbv = counts_b ? min(max(counts_b[f], 0), nb) : nb;
    const int a0 = blockIdx.y * 16, b0 = blockIdx.x * 16;
    const int t = threadIdx.y * 16 + threadIdx.x;
    if (t < 16) { if (a0 + t < na) sa[t] = make_box3(A0 + (size_t)(a0 + t) * 7); }
    else if (t < 32) { if (b0 + t - 16 < nbv) sb[t - 16] = make_box3(B0 + (size_t)(b0 + t - 16) * 7); }
    __syncthreads();
    const int ai = a0 + threadIdx.y, bi = b0 + threadIdx.x;
    if (ai >= na || bi >= nb) return;
    float iou = 0.f;
    if (bi < nbv) {
        const Box3& A = sa[threadIdx.y];
        const Box3& B = sb[threadIdx.x];
        const float ov_bev = rbox_overlap(A.bev, B.bev);
        const float hmin = fmaxf(A.cy - A.h, B.cy - B.h), hmax = fminf(A.cy, B.cy);
        const float ov3 = ov_bev * fmaxf(hmax - hmin, 0.f);
        iou = ov3 / fmaxf(A.vol + B.vol - ov3, 1e-7f);
    }
    iou_out[((size_t)f * na + ai) * nb + bi] = iou;
}
}  // namespace jm

extern "C" int jm_boxes_iou3d_batched(int batch, int num_a, const float* boxes_a, int num_b, const float* boxes_b,
                                      const int* counts_b, float* iou3d, jm_stream_t stream) {
    JM_REQUIRE(batch >= 0 && num_a >= 0 && num_b >= 0 && batch <= 65535, "boxes_iou3d_batched: bad sizes");
    if (batch == 0 || num_a == 0 || num_b == 0) return JM_OK;
    JM_REQUIRE(boxes_a && boxes_b && iou3d, "boxes_iou3d_batched: null pointer");
    hipLaunchKernelGGL(iou3d_batched_kernel, dim3(divup(num_b, 16), divup(num_a, 16), batch), dim3(16, 16), 0,
                       (hipStream_t)stream, num_a, boxes_a, num_b, boxes_b, counts_b, iou3d);
    return check_launch("boxes_iou3d_batched");
}

extern "C" size_t jm_nms_workspace_bytes(int boxes_num) {
    if (boxes_num <= 0) return 0;
    const size_t cb = (size_t)(boxes_num + 63) / 64;
    return (size_t)boxes_num * cb * sizeof(unsigned long long);
}

static int launch_nms_mask(int nprob, int nmax, const int* counts, const float* boxes, float thresh, int normal,
                           unsigned long long* mask, hipStream_t s) {
    const long long cb = (nmax + 63) / 64;
    const long long tiles = cb * (cb + 1) / 2;
    JM_REQUIRE(tiles < (1LL << 31) && nprob <= 65535, "nms_mask: too many boxes / problems");
    if (nmax <= NMS_LANE_MAX) {                                    // latency-bound sizes: one pair per lane
        const dim3 lgrid((unsigned)divup((long long)nmax * cb, 4), (unsigned)nprob);
        if (normal)
            hipLaunchKernelGGL(nms_mask_lane_kernel<true>, lgrid, dim3(256), 0, s, nmax, counts, (int)cb, thresh, boxes, mask);
        else
            hipLaunchKernelGGL(nms_mask_lane_kernel<false>, lgrid, dim3(256), 0, s, nmax, counts, (int)cb, thresh, boxes, mask);
        return check_launch("nms_mask (lane)");
    }
    dim3 grid((unsigned)tiles, (unsigned)nprob);
    if (normal)
        hipLaunchKernelGGL(nms_mask_kernel<true>, grid, dim3(64), 0, s, nmax, counts, (int)cb, thresh, boxes, mask);
    else
        hipLaunchKernelGGL(nms_mask_kernel<false>, grid, dim3(64), 0, s, nmax, counts, (int)cb, thresh, boxes, mask);
    return check_launch("nms_mask");
}

static int launch_nms_reduce(int nprob, int nmax, const int* counts, const unsigned long long* mask, int64_t* keep,
                             int* num_keep, hipStream_t s) {
    const int cb = (nmax + 63) / 64;
    const size_t lds = (size_t)(cb + 1) * sizeof(unsigned long long);
    JM_REQUIRE(lds <= 160 * 1024, "nms: %d boxes exceed the on-device reduce capacity", nmax);
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)nms_reduce_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(nms_reduce_kernel, dim3(nprob), dim3(NMS_RT), lds, s, nmax, counts, cb, mask, (long long*)keep,
                       num_keep);
    return check_launch("nms_reduce");
}

extern "C" int jm_nms_mask(int boxes_num, const float* boxes, float nms_overlap_thresh, int normal,
                           unsigned long long* mask, jm_stream_t stream) {
    JM_REQUIRE(boxes_num >= 0, "nms_mask: bad size");
    if (boxes_num == 0) return JM_OK;
    JM_REQUIRE(boxes && mask, "nms_mask: null pointer");
    return launch_nms_mask(1, boxes_num, nullptr, boxes, nms_overlap_thresh, normal, mask, (hipStream_t)stream);
}

extern "C" int jm_nms(int boxes_num, const float* boxes, float nms_overlap_thresh, int normal, int64_t* keep,
                      int* num_keep, void* ws, size_t ws_bytes, jm_stream_t stream) {
    JM_REQUIRE(boxes_num >= 0, "nms: bad size");
    JM_REQUIRE(num_keep, "nms: null num_keep");
    if (boxes_num == 0) {
        (void)jm_zero_async(num_keep, sizeof(int), (hipStream_t)stream);
        return check_launch("nms(memset)");
    }
    JM_REQUIRE(boxes && keep && ws, "nms: null pointer");
    if (ws_bytes < jm_nms_workspace_bytes(boxes_num)) {
        set_error("nms: workspace %zu < %zu bytes", ws_bytes, jm_nms_workspace_bytes(boxes_num));
        return JM_EWORKSPACE;
    }
    int rc = launch_nms_mask(1, boxes_num, nullptr, boxes, nms_overlap_thresh, normal, (unsigned long long*)ws, (hipStream_t)stream);
    if (rc) return rc;
    return launch_nms_reduce(1, boxes_num, nullptr, (const unsigned long long*)ws, keep, num_keep, (hipStream_t)stream);
}

namespace jm {
// mask + reduce for nprob problems; shared with proposal.hip
int launch_nms_batched(int nprob, int nmax, const int* counts, const float* boxes, float thresh, int normal,
                       int64_t* keep, int* num_keep, void* mask_ws, hipStream_t s) {
    int rc = launch_nms_mask(nprob, nmax, counts, boxes, thresh, normal, (unsigned long long*)mask_ws, s);
    if (rc) return rc;
    return launch_nms_reduce(nprob, nmax, counts, (const unsigned long long*)mask_ws, keep, num_keep, s);
}
}  // namespace jm

extern "C" int jm_nms_normal_first_k_batched(int num_problems, int max_boxes, const int* counts, const float* boxes,
                                            float nms_overlap_thresh, int first_k, int64_t* keep, int* num_keep,
                                            jm_stream_t stream) {
    JM_REQUIRE(num_problems >= 0 && max_boxes >= 0 && first_k >= 0, "nms_normal_first_k: bad sizes");
    if (num_problems == 0) return JM_OK;
    JM_REQUIRE(num_keep, "nms_normal_first_k: null num_keep");
    if (max_boxes == 0 || first_k == 0) {
        (void)jm_zero_async(num_keep, sizeof(int) * num_problems, (hipStream_t)stream);
        return check_launch("nms_normal_first_k(memset)");
    }
    JM_REQUIRE(boxes && keep, "nms_normal_first_k: null pointer");
    return launch_nms_first_k(num_problems, max_boxes, counts, boxes, nms_overlap_thresh, 1, first_k, first_k, keep, num_keep,
                              nullptr, (hipStream_t)stream);
}

extern "C" int jm_nms_batched(int num_problems, int max_boxes, const int* counts, const float* boxes,
                              float nms_overlap_thresh, int normal, int64_t* keep, int* num_keep, void* ws,
                              size_t ws_bytes, jm_stream_t stream) {
    JM_REQUIRE(num_problems >= 0 && max_boxes >= 0, "nms_batched: bad sizes");
    if (num_problems == 0) return JM_OK;
    JM_REQUIRE(num_keep, "nms_batched: null num_keep");
    if (max_boxes == 0) {
        (void)jm_zero_async(num_keep, sizeof(int) * num_problems, (hipStream_t)stream);
        return check_launch("nms_batched(memset)");
    }
    JM_REQUIRE(counts && boxes && keep && ws, "nms_batched: null pointer");
    const size_t need = jm_nms_workspace_bytes(max_boxes) * (size_t)num_problems;
    if (ws_bytes < need) { set_error("nms_batched: workspace %zu < %zu bytes", ws_bytes, need); return JM_EWORKSPACE; }
    int rc = launch_nms_mask(num_problems, max_boxes, counts, boxes, nms_overlap_thresh, normal, (unsigned long long*)ws,
                             (hipStream_t)stream);
    if (rc) return rc;
    return launch_nms_reduce(num_problems, max_boxes, counts, (const unsigned long long*)ws, keep, num_keep,
                             (hipStream_t)stream);
}
