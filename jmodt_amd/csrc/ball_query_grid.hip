// ball_query_grid.hip — radius neighbour search through a per-frame hash grid, BIT-EXACT with the sequential scan of
// ball_query_kernel_fast (jmodt/ops/pointnet2/src/ball_query_gpu.cu:9-45): the first `nsample` point indices, ASCENDING,
// with d2(centre, point) < r*r, back-filled with the first hit; no hit leaves the caller's fill.
//
// The brute-force kernel (ball_query.hip) evaluates all n*m pairs: 537 M distance evaluations at (16384, 4096, B = 8) where
// a centre's ball holds a handful of points.  Here:
//   1. bq_grid_build_kernel — one workgroup per frame bins the frame's points into a hash grid of <= 32768 buckets (cell edge
//      ~ the largest radius, bucket = hash of the integer cell coordinates: no bounding box, no dependence on the extent)
//      entirely in LDS (histogram with ds_add, exclusive scan, scatter with ds_add cursors) and writes the points bucket-
//      sorted as float4 {x, y, z, index} + a (start, end) pair per bucket.
//   2. bq_grid_query_kernel — one WAVE per centre: the lanes look up the buckets of the <= 64 cells the ball can touch,
//      flatten their point ranges with a wave prefix sum, evaluate the candidates 64 at a time with EXACTLY the reference's
//      distance expression (sqdist3: same contraction as every other kernel here), and set bit `index` of a per-wave LDS
//      bitmap per hit.  The bitmap IS the ascending-index order: the output row is its first nsample set bits (popcount +
//      wave prefix sum), whatever order cells, buckets and hash collisions delivered the candidates in — a point of another
//      cell that collides into a visited bucket is just one more candidate that fails or passes the same distance test, and
//      a bucket visited twice sets the same bits twice.
// Exactness of the candidate set rests on two facts only: (a) the cell coordinate floor(fl(x * inv_h)) is a monotone function
// of x, so every point with |p - c| <= rp (per axis) lies in a cell of the range [cell(c - rp), cell(c + rp)]; (b) a hit
// (fl-evaluated d2 < fl(r*r)) has |p - c| <= r (1 + 2e-6) per axis, far inside the padded rp = r (1 + 2^-10) + 2^-18 (|c| + 1).
// Centres whose range would exceed 64 cells (pathological coordinates) scan the frame's whole sorted array instead.
#include "jm_grid.h"

namespace jm {

// ------------------------------------------------------------------ query: one wave per centre
struct BgParams {
    int n, m, b;
    int gx;                 // centre groups per frame (workgroups of BG_WAVES waves x cpw centres)
    int cpw;                // centres per wave (<= 4)
    int words;              // bitmap words per radius (ceil(n / 32) rounded up to 256)
    int T;                  // buckets per frame (power of two)
    float inv_h;
    float r[2], r2[2];
    int ns[2];
    int* idx[2];
    unsigned long long* evals;   // distance evaluations of the launch (one atomic per wave): the op's real work, for the profile
};

constexpr int BG_WAVES = 4;
constexpr int BG_CPW = 4;
constexpr int BG_L2 = 128;               // second-level words per bitmap (covers 128 * 32 * 32 = 131072 points)

// the cell range one centre's padded reach touches + this lane's bucket range
struct BgCells {
    int cnt, start;         // this lane's candidate range in the sorted array
};

template <int NR>
__global__ void __launch_bounds__(64 * BG_WAVES)
bq_grid_query_kernel(BgParams p, const float* __restrict__ new_xyz, const uint2* __restrict__ tbl, const float4* __restrict__ sorted) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];        // [BG_WAVES][NR][words]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // whole frames per XCD (workgroup g runs on XCD g % 8): a frame's table and sorted points stay in one L2
    const int slot = blockIdx.x >> 3;
    const int bi = (blockIdx.x & 7) + 8 * (slot / p.gx);
    const int grp = slot % p.gx;
    if (bi >= p.b) return;
    // per radius: `words` bitmap words (bit k = point k is a hit) + a second level of words / 32 words (bit j = bitmap word j
    // is non-zero; used for clouds of more than 16384 points, where reading every bitmap word per centre would dominate)
    const int stride = p.words + BG_L2;
    unsigned* bm = lds + (size_t)wave * NR * stride;
    const int wpl = p.words >> 6;                       // bitmap words per lane (words is a multiple of 256: wpl % 4 == 0)
    for (int r = 0; r < NR; ++r) {
        for (int w = 0; w < wpl; w += 4) *reinterpret_cast<uint4*>(bm + r * stride + lane * wpl + w) = make_uint4(0u, 0u, 0u, 0u);
        for (int w = lane; w < BG_L2; w += 64) bm[r * stride + p.words + w] = 0u;
    }
    const uint2* tb = tbl + (size_t)bi * p.T;
    const float4* so = sorted + (size_t)bi * p.n;
    const unsigned tmask = (unsigned)p.T - 1u;
    const float rmax = NR == 2 ? fmaxf(p.r[0], p.r[1]) : p.r[0];
    const int c0 = (grp * BG_WAVES + wave) * p.cpw;                   // this wave's first centre (wave-uniform)
    if (c0 >= p.m) return;
    const int nc = min(p.cpw, p.m - c0);
    // the coordinates of the wave's (<= 4, consecutive) centres: one load
    const float cv = new_xyz[((size_t)bi * p.m + c0) * 3 + min(lane, nc * 3 - 1)];
    auto coord = [&](int ci, int a) { return __builtin_amdgcn_readlane(__float_as_int(cv), ci * 3 + a); };

    // bucket range of this lane's cell for centre ci (issued one centre ahead of its use)
    auto lookup = [&](int ci) {
        BgCells out{0, 0};
        const float cx = __int_as_float(coord(ci, 0)), cy = __int_as_float(coord(ci, 1)), cz = __int_as_float(coord(ci, 2));
        // padded per-axis reach and the cell range it touches
        const float rpx = rmax * 1.0009765625f + 3.8146973e-6f * (fabsf(cx) + 1.f);
        const float rpy = rmax * 1.0009765625f + 3.8146973e-6f * (fabsf(cy) + 1.f);
        const float rpz = rmax * 1.0009765625f + 3.8146973e-6f * (fabsf(cz) + 1.f);
        const int x0 = bg_cell(cx - rpx, p.inv_h), x1 = bg_cell(cx + rpx, p.inv_h);
        const int y0 = bg_cell(cy - rpy, p.inv_h), y1 = bg_cell(cy + rpy, p.inv_h);
        const int z0 = bg_cell(cz - rpz, p.inv_h), z1 = bg_cell(cz + rpz, p.inv_h);
        const long long nx = (long long)x1 - x0 + 1, ny = (long long)y1 - y0 + 1, nz = (long long)z1 - z0 + 1;
        const bool whole = !(nx >= 1 && ny >= 1 && nz >= 1 && nx * ny * nz <= 64);     // NaN / far-out centres: scan everything
        if (whole) {
            if (lane == 0) out.cnt = p.n;
        } else if (lane < (int)(nx * ny * nz)) {
            // lane -> (ax, ay, az), az fastest; (lane + 0.5) / nz is never within 1/128 of an integer: the float quotient's floor is exact
            const int t = (int)(((float)lane + 0.5f) * __builtin_amdgcn_rcpf((float)(int)nz));
            const int az = lane - t * (int)nz;
            const int ax = (int)(((float)t + 0.5f) * __builtin_amdgcn_rcpf((float)(int)ny));
            const int ay = t - ax * (int)ny;
            const uint2 se = tb[bg_bucket(x0 + ax, y0 + ay, z0 + az, tmask)];
            out.start = (int)se.x;
            out.cnt = (int)(se.y - se.x);
        }
        return out;
    };

    BgCells nxt = lookup(0);
    int evaluated = 0;
    for (int ci = 0; ci < nc; ++ci) {
        const BgCells cur = nxt;
        if (ci + 1 < nc) nxt = lookup(ci + 1);                          // its table loads fly under this centre's work
        const int c = c0 + ci;
        const float cx = __int_as_float(coord(ci, 0)), cy = __int_as_float(coord(ci, 1)), cz = __int_as_float(coord(ci, 2));
        const int incl = wave_incl_scan_i32_dpp(cur.cnt);
        const int excl = incl - cur.cnt;
        const int total = __builtin_amdgcn_readlane(incl, 63);
        const int rel = cur.start - excl;                // candidate t of this lane's range lives at sorted[rel + t]
        evaluated += total;
        for (int t0 = 0; t0 < total; t0 += 64) {
            const int t = t0 + lane;
            const bool valid = t < total;
            int lo = 0, hi = 63;                         // last lane j with excl_j <= t (its range holds candidate t)
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                const int mid = (lo + hi + 1) >> 1;
                const int e = __shfl(excl, mid);
                const bool le = e <= t;
                lo = le ? mid : lo;
                hi = le ? hi : mid - 1;
            }
            const int addr = __shfl(rel, lo) + t;
            const float4 q = so[valid ? addr : 0];
            const float d2 = sqdist3(cx - q.x, cy - q.y, cz - q.z);     // (new - x), ball_query_gpu.cu:33
            const int k = __float_as_int(q.w);
#pragma unroll
            for (int r = 0; r < NR; ++r)
                if (valid && d2 < p.r2[r]) {
                    atomicOr(&bm[r * stride + (k >> 5)], 1u << (k & 31));
                    if (wpl > 8) atomicOr(&bm[r * stride + p.words + (k >> 10)], 1u << ((k >> 5) & 31));
                }
        }
        __threadfence_block();      // the wave's ds_or traffic is complete before the bitmap is read back
        // the first ns set bits of each bitmap, ascending = the reference's scan order; back-fill with the first hit
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            unsigned* row = bm + r * stride + lane * wpl;
            int pc = 0;
            if (wpl <= 8) {                        // n <= 16384 (8 words per lane) or n <= 8192 (4): this lane's bits in registers
                const uint4 lo4 = *reinterpret_cast<const uint4*>(row);
                const uint4 hi4 = wpl == 8 ? *reinterpret_cast<const uint4*>(row + 4) : make_uint4(0u, 0u, 0u, 0u);
                const unsigned wd[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
#pragma unroll
                for (int w = 0; w < 8; ++w) pc += __popc(wd[w]);
                const int pin = wave_incl_scan_i32_dpp(pc);
                const int hits = __builtin_amdgcn_readlane(pin, 63);
                if (hits > 0) {                                           // wave-uniform
                    const int ns = p.ns[r];
                    int* out = p.idx[r] + ((size_t)bi * p.m + c) * ns;
                    int first_here = -1;
                    if (pc > 0) {
                        int rank = pin - pc;
#pragma unroll
                        for (int w = 0; w < 8; ++w) {
                            unsigned bits = wd[w];
                            if (bits && first_here < 0) first_here = (lane * wpl + w) * 32 + __ffs((int)bits) - 1;
                            while (bits && rank < ns) {
                                out[rank++] = (lane * wpl + w) * 32 + __ffs((int)bits) - 1;
                                bits &= bits - 1;
                            }
                        }
                        *reinterpret_cast<uint4*>(row) = make_uint4(0u, 0u, 0u, 0u);
                        if (wpl == 8) *reinterpret_cast<uint4*>(row + 4) = make_uint4(0u, 0u, 0u, 0u);
                    }
                    const unsigned long long have = __ballot(pc > 0);
                    const int first = __builtin_amdgcn_readlane(first_here, __ffsll((long long)have) - 1);
                    for (int s2 = hits + lane; s2 < ns; s2 += 64) out[s2] = first;
                } else {                                                  // no hit: 0, what the reference's caller pre-fills
                    int* out = p.idx[r] + ((size_t)bi * p.m + c) * p.ns[r];
                    for (int s2 = lane; s2 < p.ns[r]; s2 += 64) out[s2] = 0;
                }
            } else {
                // only the bitmap words flagged in the second level are read: this lane owns words [lane * wpl, (lane + 1) * wpl)
                unsigned* l2 = bm + r * stride + p.words;
                const int w_lo = lane * wpl, w_hi = w_lo + wpl;
                auto next_flagged = [&](int w) {               // first flagged bitmap word index >= w (or w_hi)
                    while (w < w_hi) {
                        const unsigned f = l2[w >> 5] >> (w & 31);
                        if (f) return w + __ffs((int)f) - 1;
                        w = (w | 31) + 1;
                    }
                    return w_hi;
                };
                for (int w = next_flagged(w_lo); w < w_hi; w = next_flagged(w + 1)) pc += __popc(bm[r * stride + w]);
                const int pin = wave_incl_scan_i32_dpp(pc);
                const int hits = __builtin_amdgcn_readlane(pin, 63);
                if (hits > 0) {                                           // wave-uniform
                    const int ns = p.ns[r];
                    int* out = p.idx[r] + ((size_t)bi * p.m + c) * ns;
                    int rank = pin - pc;
                    int first_here = -1;
                    if (pc > 0) {
                        for (int w = next_flagged(w_lo); w < w_hi; w = next_flagged(w + 1)) {
                            unsigned bits = bm[r * stride + w];
                            if (first_here < 0) first_here = w * 32 + __ffs((int)bits) - 1;
                            while (bits && rank < ns) {
                                out[rank++] = w * 32 + __ffs((int)bits) - 1;
                                bits &= bits - 1;
                            }
                            bm[r * stride + w] = 0u;
                        }
                    }
                    const unsigned long long have = __ballot(pc > 0);
                    const int first = __builtin_amdgcn_readlane(first_here, __ffsll((long long)have) - 1);
                    for (int s2 = hits + lane; s2 < ns; s2 += 64) out[s2] = first;
                    __threadfence_block();
                    for (int w = lane; w < BG_L2; w += 64) l2[w] = 0u;    // (every flagged word was cleared by its owner above)
                } else {
                    int* out = p.idx[r] + ((size_t)bi * p.m + c) * p.ns[r];
                    for (int s2 = lane; s2 < p.ns[r]; s2 += 64) out[s2] = 0;
                }
            }
        }
    }
    if (p.evals && lane == 0) atomicAdd(p.evals + (blockIdx.x & 31), (unsigned long long)evaluated);   // 32 slots: no hot address
}

static int bg_words(int n) { return divup(divup(n, 32), 256) * 256; }

static bool bg_applies(int n) { return n >= BG_MIN_N && n <= BG_MAX_N; }
static bool bg_radius_ok(float r) { return r > 0.f && r < 1e30f; }          // (false for NaN): the brute-force kernel takes the rest

static int bg_table_size(int n) {
    int t = 4096;
    while (t < n && t < BG_T_MAX) t *= 2;          // load factor <= 1 (2n buckets: +4 us of build for -1 us of query at 16384 points)
    return t;
}

struct BgWs { uint2* tbl; float4* sorted; unsigned long long* evals; };

static BgWs bg_carve(int b, int n, void* ws) {
    BgWs w;
    w.tbl = reinterpret_cast<uint2*>(ws);
    w.sorted = reinterpret_cast<float4*>(reinterpret_cast<char*>(ws) + align_up((size_t)b * BG_T_MAX * sizeof(uint2), 256));
    w.evals = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(w.sorted) + align_up((size_t)b * n * sizeof(float4), 256));
    return w;
}

// step 1: bin the frames' points (depends on xyz and the cell radius only — not on the centres)
static int bg_build(int b, int n, float cell_radius, const float* xyz, void* ws, hipStream_t s) {
    const float h = cell_radius * 1.01f;
    const int T = bg_table_size(n);
    const BgWs w = bg_carve(b, n, ws);
#define JM_BG_BUILD(TPT, PPT)                                                                                                   \
    do {                                                                                                                       \
        (void)hipFuncSetAttribute((const void*)bq_grid_build_kernel<TPT, PPT>, hipFuncAttributeMaxDynamicSharedMemorySize,     \
                                  1024 * TPT * 4);                                                                             \
        hipLaunchKernelGGL((bq_grid_build_kernel<TPT, PPT>), dim3((unsigned)b), dim3(1024), 1024 * TPT * 4, s, n, 1.f / h, xyz, \
                           w.tbl, w.sorted);                                                                                   \
    } while (0)
    if (T == 4096) JM_BG_BUILD(4, 16);              // n <= 4096
    else if (T == 8192) JM_BG_BUILD(8, 16);
    else if (T == 16384) JM_BG_BUILD(16, 16);       // n <= 16384
    else JM_BG_BUILD(32, 0);
#undef JM_BG_BUILD
    return check_launch("ball_query(grid build)");
}

// step 2: the searches on a grid built with `cell_radius` (any positive value is CORRECT: the cell range of a ball is computed
// from the ball's own radius; a cell radius close to the largest search radius is the fast one)
static int bg_query(int b, int n, int m, float cell_radius, int nr, const float* radius, const int* nsample, const float* new_xyz,
                    int* const* idx, void* ws, hipStream_t s) {
    const float h = cell_radius * 1.01f;
    const BgWs w = bg_carve(b, n, ws);
    (void)jm_zero_async(w.evals, 32 * sizeof(unsigned long long), s);
    BgParams p{};
    p.n = n; p.m = m; p.b = b; p.inv_h = 1.f / h; p.T = bg_table_size(n);
    p.evals = w.evals;
    p.words = bg_words(n);
    p.cpw = 1;                                   // centres per wave: as few as keeps >= ~2048 workgroups in the launch
    while (p.cpw < BG_CPW && (long long)b * m / (BG_WAVES * p.cpw * 2) >= 2048) p.cpw *= 2;
    p.gx = divup(m, BG_WAVES * p.cpw);
    for (int r = 0; r < 2; ++r) {
        const int rr = r < nr ? r : 0;
        p.r[r] = radius[rr]; p.r2[r] = radius[rr] * radius[rr];     // float product, as ball_query_gpu.cu:24
        p.ns[r] = nsample[rr]; p.idx[r] = idx[rr];
    }
    const size_t lds = (size_t)BG_WAVES * nr * (p.words + BG_L2) * sizeof(unsigned);
    const long long groups = (long long)divup(b, 8) * 8 * p.gx;
    JM_REQUIRE(groups < (1LL << 31), "ball_query: too many centres");
    if (nr == 1) {
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)bq_grid_query_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((bq_grid_query_kernel<1>), dim3((unsigned)groups), dim3(64 * BG_WAVES), lds, s, p, new_xyz, w.tbl, w.sorted);
    } else {
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)bq_grid_query_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((bq_grid_query_kernel<2>), dim3((unsigned)groups), dim3(64 * BG_WAVES), lds, s, p, new_xyz, w.tbl, w.sorted);
    }
    return check_launch("ball_query(grid)");
}

static int launch_ball_query_grid(int b, int n, int m, int nr, const float* radius, const int* nsample, const float* new_xyz,
                                  const float* xyz, int* const* idx, void* ws, hipStream_t s) {
    float rmax = radius[0];
    for (int r = 1; r < nr; ++r) rmax = fmaxf(rmax, radius[r]);
    const int rc = bg_build(b, n, rmax, xyz, ws, s);
    return rc ? rc : bg_query(b, n, m, rmax, nr, radius, nsample, new_xyz, idx, ws, s);
}

}  // namespace jm

using namespace jm;

extern "C" size_t jm_ball_query_workspace_bytes(int b, int n) {
    if (b < 1 || !bg_applies(n)) return 0;
    return align_up((size_t)b * BG_T_MAX * sizeof(uint2), 256) + align_up((size_t)b * n * sizeof(float4), 256) + 256;
}

/* byte offset of the launch's distance-evaluation counters (32 x uint64, to be summed) inside the workspace: the number of
 * (centre, candidate) pairs the last call on that workspace evaluated, readable after the stream has finished (bench.py
 * reports it as evals/s) */
extern "C" size_t jm_ball_query_evals_offset(int b, int n) {
    if (b < 1 || !bg_applies(n)) return 0;
    return align_up((size_t)b * BG_T_MAX * sizeof(uint2), 256) + align_up((size_t)b * n * sizeof(float4), 256);
}

extern "C" int jm_ball_query_ws(int b, int n, int m, float radius, int nsample, const float* new_xyz, const float* xyz, int* idx,
                                void* ws, size_t ws_bytes, jm_stream_t stream) {
    const size_t need = jm_ball_query_workspace_bytes(b, n);
    if (!ws || need == 0 || m == 0 || !bg_radius_ok(radius)) return jm_ball_query(b, n, m, radius, nsample, new_xyz, xyz, idx, stream);
    JM_REQUIRE(b >= 0 && m >= 0 && nsample >= 1 && nsample <= 1024, "ball_query: bad sizes");
    JM_REQUIRE(new_xyz && xyz && idx, "ball_query: null pointer");
    if (ws_bytes < need) { set_error("ball_query: workspace %zu < %zu bytes", ws_bytes, need); return JM_EWORKSPACE; }
    JM_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 15u) == 0, "ball_query: workspace must be 16-byte aligned");
    int* idxs[2] = {idx, nullptr};
    return launch_ball_query_grid(b, n, m, 1, &radius, &nsample, new_xyz, xyz, idxs, ws, (hipStream_t)stream);
}

extern "C" int jm_ball_query_dual_ws(int b, int n, int m, float radius0, int nsample0, float radius1, int nsample1,
                                     const float* new_xyz, const float* xyz, int* idx0, int* idx1, void* ws, size_t ws_bytes,
                                     jm_stream_t stream) {
    const size_t need = jm_ball_query_workspace_bytes(b, n);
    if (!ws || need == 0 || m == 0 || !bg_radius_ok(radius0) || !bg_radius_ok(radius1))
        return jm_ball_query_dual(b, n, m, radius0, nsample0, radius1, nsample1, new_xyz, xyz, idx0, idx1, stream);
    JM_REQUIRE(b >= 0 && m >= 0 && nsample0 >= 1 && nsample0 <= 1024 && nsample1 >= 1 && nsample1 <= 1024, "ball_query: bad sizes");
    JM_REQUIRE(new_xyz && xyz && idx0 && idx1, "ball_query: null pointer");
    if (ws_bytes < need) { set_error("ball_query: workspace %zu < %zu bytes", ws_bytes, need); return JM_EWORKSPACE; }
    JM_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 15u) == 0, "ball_query: workspace must be 16-byte aligned");
    const float rad[2] = {radius0, radius1};
    const int ns[2] = {nsample0, nsample1};
    int* idxs[2] = {idx0, idx1};
    return launch_ball_query_grid(b, n, m, 2, rad, ns, new_xyz, xyz, idxs, ws, (hipStream_t)stream);
}

/* the two steps of the *_ws entries on their own: the BUILD depends on the points and the cell radius only, so a caller can run
 * it early / on another stream (ops/pointnet2/pyramid.py builds every level's grid on the FPS side stream, behind the sampling
 * that produces the level's points) and leave only the query on the critical path.  cell_radius: the largest radius the grid
 * will be searched with (any positive value gives correct results).  radius1 <= 0 or idx1 == NULL: one radius. */
extern "C" int jm_ball_query_grid_build(int b, int n, float cell_radius, const float* xyz, void* ws, size_t ws_bytes, jm_stream_t stream) {
    const size_t need = jm_ball_query_workspace_bytes(b, n);
    JM_REQUIRE(need > 0 && bg_radius_ok(cell_radius), "ball_query_grid_build: no grid form for n = %d / radius %g", n, (double)cell_radius);
    JM_REQUIRE(xyz && ws && (reinterpret_cast<uintptr_t>(ws) & 15u) == 0, "ball_query_grid_build: null / unaligned pointer");
    if (ws_bytes < need) { set_error("ball_query_grid_build: workspace %zu < %zu bytes", ws_bytes, need); return JM_EWORKSPACE; }
    return bg_build(b, n, cell_radius, xyz, ws, (hipStream_t)stream);
}

extern "C" int jm_ball_query_grid_query(int b, int n, int m, float cell_radius, float radius0, int nsample0, float radius1, int nsample1,
                                        const float* new_xyz, int* idx0, int* idx1, void* ws, size_t ws_bytes, jm_stream_t stream) {
    const size_t need = jm_ball_query_workspace_bytes(b, n);
    JM_REQUIRE(need > 0 && bg_radius_ok(cell_radius) && bg_radius_ok(radius0), "ball_query_grid_query: no grid form for these arguments");
    if (m == 0) return JM_OK;
    const int nr = (idx1 && radius1 > 0.f) ? 2 : 1;
    JM_REQUIRE(nr == 1 || bg_radius_ok(radius1), "ball_query_grid_query: bad second radius");
    JM_REQUIRE(m >= 0 && nsample0 >= 1 && nsample0 <= 1024 && (nr == 1 || (nsample1 >= 1 && nsample1 <= 1024)), "ball_query_grid_query: bad sizes");
    JM_REQUIRE(new_xyz && idx0 && ws && (reinterpret_cast<uintptr_t>(ws) & 15u) == 0, "ball_query_grid_query: null / unaligned pointer");
    if (ws_bytes < need) { set_error("ball_query_grid_query: workspace %zu < %zu bytes", ws_bytes, need); return JM_EWORKSPACE; }
    const float rad[2] = {radius0, radius1};
    const int ns[2] = {nsample0, nsample1};
    int* idxs[2] = {idx0, idx1};
    return bg_query(b, n, m, cell_radius, nr, rad, ns, new_xyz, idxs, ws, (hipStream_t)stream);
}
