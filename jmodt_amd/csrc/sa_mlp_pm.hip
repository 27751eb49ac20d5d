// sa_mlp_pm.hip — the pre-projected set-abstraction block (hoisted first layer, see sa_mlp.hip / jm_sa_mlp_forward_pre)
// with TWO MFMA waves per SIMD and row-major activation tiles (gfx950):
//     relu(u[idx] - W1x . centre)  ->  [Conv + BN(folded) + ReLU]  ->  [Conv + BN(folded)] -> max over nsample -> ReLU
// i.e. layers 2..3 of a 3-layer SharedMLP + the max-pool of _PointnetSAModuleBase.forward
// (jmodt/ops/pointnet2/pointnet2_modules.py:46-52) on the (centre, sample) rows that QueryAndGroup would materialise.
//
// Why a second kernel (measured, DESIGN.md §4 "where sa_mlp_kernel loses its 37 %"): with ONE MFMA wave per SIMD every
// instruction next to the MFMAs (operand loads, relu(u - v), epilogues) costs matrix-pipe time — 96 / 82 cycles per MFMA
// in the two layers of sa_mlp_kernel.  Here:
//   * 8 MFMA waves per workgroup (two per SIMD, each a 32-row x 64-column block, 2 accumulators): what one wave issues
//     next to its MFMAs runs under the other wave's MFMAs (bare model tools/mfma_v2_probe.hip: 133 TF vs 119);
//   * activations ROW-major in LDS ([row][K + 4], stride / 4 odd: conflict-free): a lane's eight k-values of a 16-deep
//     k-tile are contiguous -> 2 ds_read_b128 per k-tile and no address arithmetic (k-major: 8 ds_read2_b32 + 8 v_add);
//     weights are packed with the matching k order by the caller (a column permutation in front of jm_sa_mlp_pack);
//   * the hidden layer is computed TRANSPOSED (A = weights, B = activations: D[channel][row]) so a lane holds 4
//     consecutive channels of its row and the epilogue is 4 ds_write_b128 per accumulator; the last layer keeps rows in
//     registers (normal orientation), so max over nsample stays an in-register reduction;
//   * u is POINT-major (B, N, C): a gathered row is C contiguous floats -> 16-byte loads and ds_write_b128 in the four
//     gather waves (16 + 16 instructions per thread and tile instead of 64 + 64); its producers write that layout
//     (conv1d_stack.hip / rcnn_lift.hip, point-major output mode).
// v_mfma_f32_32x32x2_f32: exact-f32 products (1e-4 parity with the fp32 reference path).
#include "jm_mfma.h"

namespace jm {

constexpr int PM_BM = 128;                 // rows per tile
constexpr int PM_VT = 8 * 128;             // per-centre table: <= 8 centres x <= 128 channels
constexpr int PM_PW = 256;                 // widest last layer
constexpr int PM_P = 16 * PM_PW;           // max-pool partials: 8 half row blocks x 2 lane halves x PM_PW columns

struct SaPmParams {
    int N, M, C, ns;                       // points per frame, centres per frame, hoisted width, nsample
    const float* u;                        // (B, N, C) point-major
    const float* new_xyz;                  // (B, M, 3)
    const int* idx;                        // (B, M, ns)
    const float* w1x;                      // (C, 4): xyz columns of the hoisted layer's weight
    int H, nblk1, nkt1;                    // hidden width, its 32-column blocks, k-tiles of the last layer (pad16(H) / 16)
    int cout, nblk2;                       // last layer's width and 32-column blocks
    int np1, np2;                          // pad128(H), pad128(cout): rows of the packed weights
    const float *W1, *W2;                  // packed (H x C), (cout x H), k order [16 kt + 8 lk + kk]
    const float *b1, *b2;                  // biases zero padded to np1 / np2
    float* out;                            // (B, cout, M), frame stride obs
    size_t obs;
    int S0, S1;                            // LDS row strides of the two tiles
    int tiles_per_frame, total_tiles, xcd_frames;
    const int* total_dev;                  // non-null: the number of tiles is read from device memory (<= total_tiles; the
                                           // duplicate-compacted form of sa_dedupe.hip, whose row count is data dependent)
    int groups, qfull, qmin;               // LISTED (sa_groups.hip): B * M, log2(ns), smallest class (2: row quads, 3: octets)
    int vt_floats, p_cols;                 // LISTED: floats of the region shared by the per-centre table and the partial-maximum rows; their columns
    const int* cls_count;                  // LISTED: [8] groups per class q of 2^q rows, q >= qmin (device memory)
    const int* glist;                      // LISTED: class q's group ids b * M + i at glist[q * groups ...]
#ifdef JM_TOOLS_BUILD
    int dbg;                               // tools build (JM_PM_DBG): timing experiments, wrong results
    long long* trace;                      // tools build: shader-clock stamps of workgroup 0's first MFMA wave (tools/sa_trace.py)
#endif
};

#ifdef JM_TOOLS_BUILD
#define PM_DBG(bit) (p.dbg & (bit))
#define PM_STAMP(k) do { if (p.trace && blockIdx.x == 0 && tid == 0 && it < 64) p.trace[it * 8 + (k)] = (long long)clock64(); } while (0)
#else
#define PM_DBG(bit) 0
#define PM_STAMP(k) do { } while (0)
#endif

struct PmSchedule {                        // = SaSchedule of sa_mlp.hip: whole frames per XCD when there are enough of them
    int n_local, xcd, slot, per, nwg, tpf, mode;
    __device__ PmSchedule(const SaPmParams& p) {
        nwg = gridDim.x; tpf = p.tiles_per_frame; mode = p.xcd_frames;
        xcd = blockIdx.x & 7; slot = blockIdx.x >> 3; per = nwg >> 3;
        const int total_tiles = p.total_dev ? min(*p.total_dev, p.total_tiles) : p.total_tiles;
        if (mode) {
            const int nb = total_tiles / tpf;
            const int local_tiles = ((nb - xcd + 7) >> 3) * tpf;
            n_local = local_tiles > slot ? (local_tiles - slot + per - 1) / per : 0;
        } else {
            n_local = total_tiles > (int)blockIdx.x ? (total_tiles - (int)blockIdx.x + nwg - 1) / nwg : 0;
        }
    }
    __device__ void tile(int i, int& bi, int& row0) const {
        const int t = mode ? slot + i * per : (int)blockIdx.x + i * nwg;
        bi = mode ? xcd + 8 * (t / tpf) : t / tpf;
        row0 = (t % tpf) * PM_BM;
    }
};

// one layer's k-loop for this wave's NB owned 32-column blocks (blocks j and j + 2 of the packed weight)
//   xp : this lane's activation row (+ 8 lk);  vp : this lane's row of the centre table (+ 8 lk), FIRST layer only
//   wp : this lane's packed weights of the first owned block (row 32 blk + lr, + 8 lk); the second block is 1024 floats on
//   FIRST: operand = relu(x - v), transposed product acc[j] = W . X^T (D[channel][row]); else acc[j] = X . W^T (D[row][channel])
//   w0: in = this stage's k-tile 0 weights (both blocks), loaded by the previous stage; out = k-tile 0 of the NEXT stage
//   (next_wp, two blocks 1024 floats apart), so that no stage starts on an L2 round trip
template <int NB, bool FIRST>
__device__ __forceinline__ void pm_ktiles(const float* __restrict__ xp, const float* __restrict__ vp, int nkt,
                                          const float* __restrict__ wp, size_t kt_stride, f32x16 (&acc)[2],
                                          float (&w0)[2][8], const float* __restrict__ next_wp) {
    float x0[8], x1[8], v0[8], v1[8], w1[NB][8];
    auto ld8 = [](float (&d)[8], const float* q) __attribute__((always_inline)) {
        const float4 lo = *reinterpret_cast<const float4*>(q), hi = *reinterpret_cast<const float4*>(q + 4);
        d[0] = lo.x; d[1] = lo.y; d[2] = lo.z; d[3] = lo.w; d[4] = hi.x; d[5] = hi.y; d[6] = hi.z; d[7] = hi.w;
    };
    auto loadx = [&](float (&x)[8], float (&v)[8], int kt) __attribute__((always_inline)) {
        ld8(x, xp + kt * 16);
        if (FIRST) ld8(v, vp + kt * 16);
    };
    auto mm = [&](const float (&x)[8], const float (&v)[8], const float (&w)[2][8]) __attribute__((always_inline)) {
        float t[8];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) t[kk] = FIRST ? fmaxf(x[kk] - v[kk], 0.f) : x[kk];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
#pragma unroll
            for (int j = 0; j < NB; ++j)
                acc[j] = FIRST ? __builtin_amdgcn_mfma_f32_32x32x2f32(w[j][kk], t[kk], acc[j], 0, 0, 0)
                               : __builtin_amdgcn_mfma_f32_32x32x2f32(t[kk], w[j][kk], acc[j], 0, 0, 0);
    };
    auto mm1 = [&](const float (&x)[8], const float (&v)[8], const float (&w)[NB][8]) __attribute__((always_inline)) {
        float t[8];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) t[kk] = FIRST ? fmaxf(x[kk] - v[kk], 0.f) : x[kk];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
#pragma unroll
            for (int j = 0; j < NB; ++j)
                acc[j] = FIRST ? __builtin_amdgcn_mfma_f32_32x32x2f32(w[j][kk], t[kk], acc[j], 0, 0, 0)
                               : __builtin_amdgcn_mfma_f32_32x32x2f32(t[kk], w[j][kk], acc[j], 0, 0, 0);
    };
    loadx(x0, v0, 0);
    int kt = 0;
    for (; kt + 2 <= nkt; kt += 2) {
#pragma unroll
        for (int j = 0; j < NB; ++j) ld8(w1[j], wp + (size_t)(kt + 1) * kt_stride + j * 1024);
        loadx(x1, v1, kt + 1);
        __builtin_amdgcn_sched_barrier(0);
        mm(x0, v0, w0);
        __builtin_amdgcn_sched_barrier(0);
        // the k-tile after next — or, on the last trip, the NEXT stage's first weights (both blocks): unconditional loads
        // on a selected address
        const bool more = kt + 2 < nkt;
        const float* q = more ? wp + (size_t)(kt + 2) * kt_stride : next_wp;
        ld8(w0[0], q);
        if (NB == 2 || !more) ld8(w0[1], q + 1024);
        loadx(x0, v0, more ? kt + 2 : kt);
        __builtin_amdgcn_sched_barrier(0);
        mm1(x1, v1, w1);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (kt < nkt) {                                                 // odd tail: w0 / x0 hold k-tile nkt - 1
        mm(x0, v0, w0);
        ld8(w0[0], next_wp); ld8(w0[1], next_wp + 1024);
    }
}

// =============================================================================== MFMA role (waves 0-7)
__device__ __forceinline__ void pm_mfma_role(const SaPmParams& p, float* lds, int tid) {
    const PmSchedule sch(p);
    if (sch.n_local == 0) return;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rb = wave >> 1, cb = wave & 1;                        // 4 row blocks x 2 interleaved column-block sets
    const int lr = lane & 31, lk = lane >> 5;
    float* X0 = lds;
    float* X1 = X0 + (size_t)PM_BM * p.S0;
    float* VT = X1 + (size_t)PM_BM * p.S1;
    float* P = VT + PM_VT;                                          // [16][PM_PW]: (half row block, lane half) x column
    float* B1 = P + PM_P;                                           // the hidden layer's bias (np1 floats)
    const int C = p.C, ns = p.ns, cout = p.cout;
    const int row = rb * 32 + lr;
    const float* x0p = X0 + (size_t)row * p.S0 + lk * 8;
    const float* x1p = X1 + (size_t)row * p.S1 + lk * 8;
    const float* vtp = VT + (size_t)(row / ns) * C + lk * 8;
#ifdef JM_PM_CONST
    constexpr int nkt0 = 8, nkt1 = 8;
    constexpr size_t st1 = 2048, st2 = 2048;
#else
    const int nkt0 = C >> 4, nkt1 = p.nkt1;
    const size_t st1 = (size_t)p.np1 * 16, st2 = (size_t)p.np2 * 16;
#endif
    const size_t lane_w = (size_t)lr * 16 + lk * 8;                 // + 512 per 32-column block
    f32x16 acc[2];
    for (int e = tid; e < p.np1; e += 512) B1[e] = p.b1[e];         // read back by every tile's accumulator set-up
    // this thread's output columns' bias (ncen * cout <= 2048 outputs per tile, 512 threads)
    const int ncen = PM_BM / ns, hpc = ns >> 4;
    float ob2[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int e = tid + 512 * q; ob2[q] = e < ncen * cout ? p.b2[e / ncen] : 0.f; }
    const float* wp1 = p.W1 + (size_t)cb * 512 + lane_w;            // hidden layer: owned blocks cb, cb + 2
    const float* wp2 = p.W2 + (size_t)cb * 512 + lane_w;            // last layer: owned blocks cb, cb + 2 (, cb + 4, cb + 6)
    // the chain of first weights: every stage this wave executes ends by loading k-tile 0 of the NEXT stage it executes
    // (a wave without column blocks in a layer skips that stage, so "next" is per wave)
    const bool has_a = cb < p.nblk1, has_b = cb < p.nblk2;
    const float* after_b = has_a ? wp1 : wp2;                       // after the last pass of the last layer: the next tile
    float wpre[2][8];
    {
        const float* q0 = has_a ? wp1 : wp2;
        const float4 a = *reinterpret_cast<const float4*>(q0), b = *reinterpret_cast<const float4*>(q0 + 4);
        const float4 c = *reinterpret_cast<const float4*>(q0 + 1024), d = *reinterpret_cast<const float4*>(q0 + 1028);
        wpre[0][0] = a.x; wpre[0][1] = a.y; wpre[0][2] = a.z; wpre[0][3] = a.w; wpre[0][4] = b.x; wpre[0][5] = b.y; wpre[0][6] = b.z; wpre[0][7] = b.w;
        wpre[1][0] = c.x; wpre[1][1] = c.y; wpre[1][2] = c.z; wpre[1][3] = c.w; wpre[1][4] = d.x; wpre[1][5] = d.y; wpre[1][6] = d.z; wpre[1][7] = d.w;
    }
    lds_barrier();                                                  // B0: tile 0, its centre table and B1 are in LDS
    for (int it = 0; it < sch.n_local; ++it) {
        int bi, row0;
        sch.tile(it, bi, row0);
        PM_STAMP(0);
        // ---------------- hidden layer, transposed: acc[j][4 rq + t] = channel 32 blk + 8 rq + 4 lk + t of row `row`
        {
            const int nb = (p.nblk1 - cb + 1) >> 1;                 // owned blocks cb, cb + 2
            if (nb > 0) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float* bq = B1 + (cb + 2 * j) * 32 + 4 * lk;       // zero padded to np1 >= 32 nblk1
                    const bool own = j < nb && !PM_DBG(16);
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {
                        const float4 bv = own ? *reinterpret_cast<const float4*>(bq + 8 * rq) : make_float4(0.f, 0.f, 0.f, 0.f);
                        acc[j][4 * rq + 0] = bv.x; acc[j][4 * rq + 1] = bv.y; acc[j][4 * rq + 2] = bv.z; acc[j][4 * rq + 3] = bv.w;
                    }
                }
                const float* nxt = has_b ? wp2 : wp1;
                if (nb == 2) pm_ktiles<2, true>(x0p, vtp, nkt0, wp1, st1, acc, wpre, nxt);
                else pm_ktiles<1, true>(x0p, vtp, nkt0, wp1, st1, acc, wpre, nxt);
                PM_STAMP(1);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (j < nb && !PM_DBG(4)) {
                        float* Y = X1 + (size_t)row * p.S1 + (cb + 2 * j) * 32 + 4 * lk;
#pragma unroll
                        for (int rq = 0; rq < 4; ++rq) {
                            float4 v;
                            v.x = fmaxf(acc[j][4 * rq + 0], 0.f); v.y = fmaxf(acc[j][4 * rq + 1], 0.f);
                            v.z = fmaxf(acc[j][4 * rq + 2], 0.f); v.w = fmaxf(acc[j][4 * rq + 3], 0.f);
                            *reinterpret_cast<float4*>(Y + 8 * rq) = v;
                        }
                    }
                }
            }
        }
        PM_STAMP(2);
        lds_barrier();                                              // B1: hidden tile complete; X0 and the table are free
        PM_STAMP(3);
        // ---------------- last layer: acc[j][4 rq + t] = row 8 rq + 4 lk + t of the block, channel 32 blk + lr
        for (int b0 = cb; b0 < p.nblk2; b0 += 4) {                  // owned blocks b0, b0 + 2
            const int nb = b0 + 2 < p.nblk2 ? 2 : 1;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
            const float* wp = p.W2 + (size_t)b0 * 512 + lane_w;
            const float* nxt = b0 + 4 < p.nblk2 ? wp + 2048 : after_b;   // the next pass of this layer, or the next tile
            if (nb == 2) pm_ktiles<2, false>(x1p, nullptr, nkt1, wp, st2, acc, wpre, nxt);
            else pm_ktiles<1, false>(x1p, nullptr, nkt1, wp, st2, acc, wpre, nxt);
            PM_STAMP(4);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (j < nb && !PM_DBG(8)) {
                    float lo = acc[j][0], hi = acc[j][8];           // rows 0-15 of the block are r < 8, rows 16-31 are r >= 8
#pragma unroll
                    for (int r = 1; r < 8; ++r) { lo = fmaxf(lo, acc[j][r]); hi = fmaxf(hi, acc[j][r + 8]); }
                    // the other lane half holds the rows + 4: both halves park their maxima (no cross-lane exchange)
                    const int col = (b0 + 2 * j) * 32 + lr;         // < 32 nblk2 <= PM_PW
                    P[((2 * rb) * 2 + lk) * PM_PW + col] = lo;
                    P[((2 * rb + 1) * 2 + lk) * PM_PW + col] = hi;
                }
            }
        }
        PM_STAMP(5);
        lds_barrier();                                              // B2: partial maxima complete (and the next tile parked)
        PM_STAMP(6);
        // ---------------- max over each centre's nsample rows (nsample / 16 half blocks x 2 lane halves), + bias, ReLU (both
        // commute with max)
        {
            float* ob = p.out + (size_t)bi * p.obs + row0 / ns;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int e = tid + 512 * q;
                if (e < (PM_DBG(2) ? 0 : ncen * cout)) {
                    const int col = e / ncen, c = e - col * ncen;
                    const float* pp = P + (size_t)(c * hpc * 2) * PM_PW + col;
                    float t = pp[0];
                    for (int h = 1; h < 2 * hpc; ++h) t = fmaxf(t, pp[(size_t)h * PM_PW]);
                    ob[(size_t)col * p.M + c] = fmaxf(t + ob2[q], 0.f);
                }
            }
        }
    }
}

// =============================================================================== gather role (waves 8-11)
template <int ROUNDS>      // 16-byte pieces per row C / 4 = 2 ROUNDS: C = 128 -> 16 rounds, 64 -> 8, 32 -> 4
__device__ __forceinline__ void pm_gather_role(const SaPmParams& p, float* lds, int ltid) {
    const PmSchedule sch(p);
    if (sch.n_local == 0) return;
    const int lane = ltid & 63;
    const int gw = __builtin_amdgcn_readfirstlane(ltid >> 6);
    float* X0 = lds;
    float* VT = X0 + (size_t)PM_BM * p.S0 + (size_t)PM_BM * p.S1;
    constexpr int chunks = 2 * ROUNDS;                              // 16-byte pieces per row: 32, 16 or 8
    constexpr int rpi = 64 / chunks;                                // rows per wave instruction
    const int C = p.C;
    const int lrow = lane / chunks, piece = lane - lrow * chunks;
    typedef float f32x4 __attribute__((ext_vector_type(4)));      // (HIP's float4 struct arrays do not stay in registers)
    f32x4 g[ROUNDS];
    auto issue = [&](int i) __attribute__((always_inline)) {
        int bi, row0;
        sch.tile(i, bi, row0);
        const int* ip = p.idx + (size_t)bi * p.M * p.ns + row0 + gw * rpi + lrow;
        const float* ub = p.u + (size_t)bi * p.N * C + 4 * piece;
        int id[ROUNDS];
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) id[r] = ip[r * 4 * rpi];
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) g[r] = *reinterpret_cast<const f32x4*>(ub + (size_t)id[r] * C);
    };
    auto store = [&]() __attribute__((always_inline)) {
        float* q = X0 + (size_t)(gw * rpi + lrow) * p.S0 + 4 * piece;
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) *reinterpret_cast<f32x4*>(q + (size_t)r * 4 * rpi * p.S0) = g[r];
    };
    // v[c][k] = W1x[k] . centre_c for the tile's 128 / ns centres: consumed by the MFMA waves as relu(u - v)
    auto write_table = [&](int i) __attribute__((always_inline)) {
        int bi, row0;
        sch.tile(i, bi, row0);
        const int ncen = PM_BM / p.ns, c0 = row0 / p.ns;
        for (int e = ltid; e < ncen * C; e += 256) {
            const int c = e / C, k = e - c * C;
            const float* cp = p.new_xyz + ((size_t)bi * p.M + c0 + c) * 3;
            const float* wv = p.w1x + (size_t)k * 4;
            VT[e] = __builtin_fmaf(wv[2], cp[2], __builtin_fmaf(wv[1], cp[1], wv[0] * cp[0]));
        }
    };
    issue(0);
    write_table(0);
    store();
    lds_barrier();                                                  // B0
    for (int it = 0; it < sch.n_local; ++it) {
        const bool has_next = it + 1 < sch.n_local;
        if (has_next && !PM_DBG(1)) issue(it + 1);                  // in flight under this tile's hidden layer
        lds_barrier();                                              // B1: the MFMA waves are done with X0 and the table
        if (has_next && !PM_DBG(1)) { store(); write_table(it + 1); }
        lds_barrier();                                              // B2
    }
}

__global__ void __launch_bounds__(768)
sa_mlp_pm_kernel(SaPmParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    if (tid < 512) {
        if (PM_DBG(32)) __builtin_amdgcn_s_setprio(3);
        if (PM_DBG(64)) { if (tid >= 256) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(1); }
        pm_mfma_role(p, lds, tid);
    }
    else if (p.C == 128) pm_gather_role<16>(p, lds, tid - 512);
    else if (p.C == 64) pm_gather_role<8>(p, lds, tid - 512);
    else pm_gather_role<4>(p, lds, tid - 512);
}

constexpr int PM_QMIN_DEFAULT = 2;
// =============================================================================== LISTED mode (round 4, sa_groups.hip)
// ball_query back-fills a list of cnt < nsample hits with copies of its first hit (ball_query_gpu.cu:36-40) and the max-pool is
// idempotent: a group of class q only needs its first 2^q rows.  A 128-row tile holds 128 >> q groups of ONE class; its rows,
// the per-centre table and the output positions go through the class list (group id -> frame, centre).  Classes of 4 rows (the
// accumulator layout pools four consecutive rows inside a lane) and 8 rows (+ one exchange between the lane halves) store their
// groups' outputs straight from the accumulators; classes of 16 .. nsample rows go through the dense role's half-block partials.
// The smallest class q_min is 2 wherever the tiles, the 32-centre table (which shares its LDS region with the partial rows: the two
// are never live in the same tile) and the biases fit the 160 KB — every shape of the reference configuration; 3 otherwise.  The k-loops, their operand order and the bias / ReLU epilogues are the dense
// kernel's (pm_ktiles), so a row's value — and therefore every output — is bit-identical.

struct PmListedSchedule {
    int ts[8], total, n_local, nwg;
    __device__ PmListedSchedule(const SaPmParams& p) {
        int acc_t = 0;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            ts[c] = acc_t;
            if (c >= p.qmin && c <= p.qfull) acc_t += (int)((((long long)p.cls_count[c] << c) + PM_BM - 1) / PM_BM);
        }
        total = acc_t; nwg = gridDim.x;
        n_local = total > (int)blockIdx.x ? (total - (int)blockIdx.x + nwg - 1) / nwg : 0;
    }
    // tile i of this workgroup: class q, first slot of the class list, groups in the class
    __device__ void tile(const SaPmParams& p, int i, int& q, int& slot0, int& cnt) const {
        const int t = (int)blockIdx.x + i * nwg;
        q = p.qmin;
#pragma unroll
        for (int c = 3; c < 8; ++c)
            if (c > p.qmin && c <= p.qfull && t >= ts[c]) q = c;   // the last class that starts at or before t (empty classes lose)
        int start = 0;
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (c == q) start = ts[c];
        slot0 = (t - start) * (PM_BM >> q);
        cnt = p.cls_count[q];
    }
};

__device__ __forceinline__ void pm_mfma_role_listed(const SaPmParams& p, float* lds, int tid) {
    const PmListedSchedule sch(p);
    if (sch.n_local == 0) return;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rb = wave >> 1, cb = wave & 1;
    const int lr = lane & 31, lk = lane >> 5;
    float* X0 = lds;
    float* X1 = X0 + (size_t)PM_BM * p.S0;
    float* VT = X1 + (size_t)PM_BM * p.S1;
    // the per-centre table and the partial-maximum rows SHARE one region: a tile of class q <= 3 has a table of 32 / 16 centres
    // and no partials (its outputs leave from the accumulators), a tile of class q >= 4 a table of <= 8 centres and the
    // partials behind it.  A workgroup's tiles ascend in class, so the table the gather role writes for the NEXT tile while this
    // tile's last layer runs either belongs to a tile without partials in flight (this tile's class <= 3) or stays below them
    float* P = VT + 8 * p.C;                                        // [16][p_cols]: (half row block, lane half) x column, classes q >= 4
    const int pcols = p.p_cols;
    float* B1 = VT + p.vt_floats;                                   // the hidden layer's bias (np1 floats)
    float* B2 = B1 + p.np1;                                         // the last layer's bias (np2 floats)
    const int C = p.C, cout = p.cout;
    const int row = rb * 32 + lr;
    const float* x0p = X0 + (size_t)row * p.S0 + lk * 8;
    const float* x1p = X1 + (size_t)row * p.S1 + lk * 8;
    const int nkt0 = C >> 4, nkt1 = p.nkt1;
    const size_t st1 = (size_t)p.np1 * 16, st2 = (size_t)p.np2 * 16;
    const size_t lane_w = (size_t)lr * 16 + lk * 8;
    f32x16 acc[2];
    for (int e = tid; e < p.np1; e += 512) B1[e] = p.b1[e];
    for (int e = tid; e < p.np2; e += 512) B2[e] = p.b2[e];
    const float* wp1 = p.W1 + (size_t)cb * 512 + lane_w;
    const float* wp2 = p.W2 + (size_t)cb * 512 + lane_w;
    const bool has_a = cb < p.nblk1, has_b = cb < p.nblk2;
    const float* after_b = has_a ? wp1 : wp2;
    float wpre[2][8];
    {
        const float* q0 = has_a ? wp1 : wp2;
        const float4 a = *reinterpret_cast<const float4*>(q0), b = *reinterpret_cast<const float4*>(q0 + 4);
        const float4 c = *reinterpret_cast<const float4*>(q0 + 1024), d = *reinterpret_cast<const float4*>(q0 + 1028);
        wpre[0][0] = a.x; wpre[0][1] = a.y; wpre[0][2] = a.z; wpre[0][3] = a.w; wpre[0][4] = b.x; wpre[0][5] = b.y; wpre[0][6] = b.z; wpre[0][7] = b.w;
        wpre[1][0] = c.x; wpre[1][1] = c.y; wpre[1][2] = c.z; wpre[1][3] = c.w; wpre[1][4] = d.x; wpre[1][5] = d.y; wpre[1][6] = d.z; wpre[1][7] = d.w;
    }
    lds_barrier();                                                  // B0
    for (int it = 0; it < sch.n_local; ++it) {
        int q, slot0, cnt;
        sch.tile(p, it, q, slot0, cnt);
        const float* vtp = VT + (size_t)(row >> q) * C + lk * 8;
        // classes of 4 / 8 rows: a group's maximum forms inside the lane (+ one exchange between the lane halves) and is stored
        // straight from the accumulators — this lane's four row quads rq belong to the groups below (output offsets, -1: padding)
        long long goff[4] = {-1, -1, -1, -1};
        if (q <= 3) {
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int slot = q == 2 ? rb * 8 + 2 * rq + lk : rb * 4 + rq;
                if (slot0 + slot < cnt) {
                    const int g = p.glist[(size_t)q * p.groups + slot0 + slot];
                    goff[rq] = (long long)((size_t)(g / p.M) * p.obs + (size_t)(g % p.M));
                }
            }
        }
        // ---------------- hidden layer, transposed (as the dense role)
        {
            const int nb = (p.nblk1 - cb + 1) >> 1;
            if (nb > 0) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float* bq = B1 + (cb + 2 * j) * 32 + 4 * lk;
                    const bool own = j < nb;
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {
                        const float4 bv = own ? *reinterpret_cast<const float4*>(bq + 8 * rq) : make_float4(0.f, 0.f, 0.f, 0.f);
                        acc[j][4 * rq + 0] = bv.x; acc[j][4 * rq + 1] = bv.y; acc[j][4 * rq + 2] = bv.z; acc[j][4 * rq + 3] = bv.w;
                    }
                }
                const float* nxt = has_b ? wp2 : wp1;
                if (nb == 2) pm_ktiles<2, true>(x0p, vtp, nkt0, wp1, st1, acc, wpre, nxt);
                else pm_ktiles<1, true>(x0p, vtp, nkt0, wp1, st1, acc, wpre, nxt);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (j < nb) {
                        float* Y = X1 + (size_t)row * p.S1 + (cb + 2 * j) * 32 + 4 * lk;
#pragma unroll
                        for (int rq = 0; rq < 4; ++rq) {
                            float4 v;
                            v.x = fmaxf(acc[j][4 * rq + 0], 0.f); v.y = fmaxf(acc[j][4 * rq + 1], 0.f);
                            v.z = fmaxf(acc[j][4 * rq + 2], 0.f); v.w = fmaxf(acc[j][4 * rq + 3], 0.f);
                            *reinterpret_cast<float4*>(Y + 8 * rq) = v;
                        }
                    }
                }
            }
        }
        lds_barrier();                                              // B1
        // ---------------- last layer: acc[j][4 rq + t] = row 8 rq + 4 lk + t of the block, channel 32 blk + lr
        for (int b0 = cb; b0 < p.nblk2; b0 += 4) {
            const int nb = b0 + 2 < p.nblk2 ? 2 : 1;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
            const float* wp = p.W2 + (size_t)b0 * 512 + lane_w;
            const float* nxt = b0 + 4 < p.nblk2 ? wp + 2048 : after_b;
            if (nb == 2) pm_ktiles<2, false>(x1p, nullptr, nkt1, wp, st2, acc, wpre, nxt);
            else pm_ktiles<1, false>(x1p, nullptr, nkt1, wp, st2, acc, wpre, nxt);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (j < nb) {
                    const int col = (b0 + 2 * j) * 32 + lr;
                    if (q <= 3) {
                        const float bias = B2[col];                 // (zero padded to np2)
#pragma unroll
                        for (int rq = 0; rq < 4; ++rq) {            // rows 8 rq + 4 lk .. + 3 of the block: one quad per lane half
                            float v = fmaxf(fmaxf(acc[j][4 * rq], acc[j][4 * rq + 1]), fmaxf(acc[j][4 * rq + 2], acc[j][4 * rq + 3]));
                            if (q == 3) v = fmaxf(v, __shfl_xor(v, 32));      // octet = this quad and the other lane half's
                            if ((q == 2 || lk == 0) && goff[rq] >= 0 && col < cout)
                                p.out[(size_t)goff[rq] + (size_t)col * p.M] = fmaxf(v + bias, 0.f);   // + bias, ReLU: as the final pass
                        }
                    } else {                                        // the dense role's partials: rows 0-15 are r < 8, rows 16-31 r >= 8
                        float lo = acc[j][0], hi = acc[j][8];
#pragma unroll
                        for (int r = 1; r < 8; ++r) { lo = fmaxf(lo, acc[j][r]); hi = fmaxf(hi, acc[j][r + 8]); }
                        P[(size_t)((2 * rb) * 2 + lk) * pcols + col] = lo;
                        P[(size_t)((2 * rb + 1) * 2 + lk) * pcols + col] = hi;
                    }
                }
            }
        }
        lds_barrier();                                              // B2
        // ---------------- classes of >= 16 rows: max over the group's 2^q / 16 half blocks x 2 lane halves, + bias, ReLU (both
        // commute with max) — the dense role's final pass with the group id from the list
        if (q >= 4) {
            const int ncen = PM_BM >> q, qpg = 2 << (q - 4);
            for (int e = tid; e < ncen * cout; e += 512) {
                const int col = e / ncen, c = e - col * ncen;
                if (slot0 + c < cnt) {
                    const int g = p.glist[(size_t)q * p.groups + slot0 + c];
                    const float* pp = P + (size_t)(c * qpg) * pcols + col;
                    float t = pp[0];
                    for (int h = 1; h < qpg; ++h) t = fmaxf(t, pp[(size_t)h * pcols]);
                    p.out[(size_t)(g / p.M) * p.obs + (size_t)col * p.M + (size_t)(g % p.M)] = fmaxf(t + B2[col], 0.f);
                }
            }
        }
    }
}

template <int ROUNDS>
__device__ __forceinline__ void pm_gather_role_listed(const SaPmParams& p, float* lds, int ltid) {
    const PmListedSchedule sch(p);
    if (sch.n_local == 0) return;
    const int lane = ltid & 63;
    const int gw = __builtin_amdgcn_readfirstlane(ltid >> 6);
    float* X0 = lds;
    float* VT = X0 + (size_t)PM_BM * p.S0 + (size_t)PM_BM * p.S1;
    constexpr int chunks = 2 * ROUNDS;
    constexpr int rpi = 64 / chunks;
    const int C = p.C;
    const int lrow = lane / chunks, piece = lane - lrow * chunks;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 g[ROUNDS];
    auto issue = [&](int i) __attribute__((always_inline)) {
        int q, slot0, cnt;
        sch.tile(p, i, q, slot0, cnt);
        const int* gl = p.glist + (size_t)q * p.groups;
        const int mask = (1 << q) - 1;
        const float* src[ROUNDS];
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const int rr = gw * rpi + lrow + r * 4 * rpi;           // row of the tile
            const int slot = slot0 + (rr >> q);
            const int gi = gl[slot < cnt ? slot : slot0];           // padding rows repeat the tile's first group (never stored)
            const int id = p.idx[(size_t)gi * p.ns + (rr & mask)];
            src[r] = p.u + ((size_t)(gi / p.M) * p.N + id) * C + 4 * piece;
        }
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) g[r] = *reinterpret_cast<const f32x4*>(src[r]);
    };
    auto store = [&]() __attribute__((always_inline)) {
        float* q = X0 + (size_t)(gw * rpi + lrow) * p.S0 + 4 * piece;
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) *reinterpret_cast<f32x4*>(q + (size_t)r * 4 * rpi * p.S0) = g[r];
    };
    auto write_table = [&](int i) __attribute__((always_inline)) {
        int q, slot0, cnt;
        sch.tile(p, i, q, slot0, cnt);
        const int* gl = p.glist + (size_t)q * p.groups;
        const int ncen = PM_BM >> q;
        for (int e = ltid; e < ncen * C; e += 256) {
            const int c = e / C, k = e - c * C;
            const int gi = gl[slot0 + c < cnt ? slot0 + c : slot0];
            const float* cp = p.new_xyz + (size_t)gi * 3;
            const float* wv = p.w1x + (size_t)k * 4;
            VT[e] = __builtin_fmaf(wv[2], cp[2], __builtin_fmaf(wv[1], cp[1], wv[0] * cp[0]));
        }
    };
    issue(0);
    write_table(0);
    store();
    lds_barrier();                                                  // B0
    for (int it = 0; it < sch.n_local; ++it) {
        const bool has_next = it + 1 < sch.n_local;
        if (has_next) issue(it + 1);
        lds_barrier();                                              // B1
        if (has_next) { store(); write_table(it + 1); }
        lds_barrier();                                              // B2
    }
}

__global__ void __launch_bounds__(768)
sa_mlp_pm_listed_kernel(SaPmParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    if (tid < 512) pm_mfma_role_listed(p, lds, tid);
    else if (p.C == 128) pm_gather_role_listed<16>(p, lds, tid - 512);
    else if (p.C == 64) pm_gather_role_listed<8>(p, lds, tid - 512);
    else pm_gather_role_listed<4>(p, lds, tid - 512);
}

// floats of the region the per-centre table and the partial rows share (see pm_mfma_role_listed)
static size_t sa_pm_listed_region(int c, int cout, int qmin) {
    const size_t table = (size_t)(PM_BM >> qmin) * c, with_partials = (size_t)8 * c + (size_t)16 * pad_to(cout, 128);
    return table > with_partials ? table : with_partials;
}
static size_t sa_pm_listed_lds_bytes(int c, int h, int cout, int qmin) {
    return ((size_t)PM_BM * (c + 4) + (size_t)PM_BM * (pad_to(h, 32) + 4) + sa_pm_listed_region(c, cout, qmin) + pad_to(h, 128) +
            pad_to(cout, 128)) * sizeof(float);
}

static size_t sa_pm_lds_bytes(int c, int h) {
    return ((size_t)PM_BM * (c + 4) + (size_t)PM_BM * (pad_to(h, 32) + 4) + PM_VT + PM_P + pad_to(h, 128)) * sizeof(float);
}

}  // namespace jm

using namespace jm;

#ifdef JM_TOOLS_BUILD
static long long* g_pm_trace = nullptr;      // tools build only: the product library keeps no state
extern "C" __attribute__((visibility("default"))) void jm_tools_set_pm_trace(long long* buf) { g_pm_trace = buf; }
#endif

extern "C" int jm_sa_mlp_pm_supported(int b, int n, int m, int c, int nsample, int hidden, int cout) {
    if (b < 0 || n < 1 || m < 1 || hidden < 1 || cout < 1) return 0;
    if (c != 32 && c != 64 && c != 128) return 0;
    if (nsample != 16 && nsample != 32 && nsample != 64) return 0;
    if (((long long)m * nsample) % PM_BM) return 0;
    if (hidden > 128 || cout > PM_PW || b > (1 << 24)) return 0;
    if ((long long)b * ((long long)m * nsample / PM_BM) >= (1LL << 31)) return 0;
    return sa_pm_lds_bytes(c, hidden) <= 160 * 1024 ? 1 : 0;
}

static int sa_mlp_pm_launch(int b, int n, int m, int c, int nsample, int hidden, int cout, const float* u_point_major,
                            const float* w1x, const float* new_xyz, const int* idx, const float* w_hidden,
                            const float* b_hidden, const float* w_out, const float* b_out, float* out, size_t obs, const int* total_dev,
                            jm_stream_t stream) {
    JM_REQUIRE(b >= 0 && m >= 0, "sa_mlp_pm: bad sizes");
    JM_REQUIRE(obs == 0 || obs >= (size_t)cout * (size_t)m, "sa_mlp_pm: output frame stride below cout * npoint");
    if (b == 0 || m == 0) return JM_OK;
    JM_REQUIRE(jm_sa_mlp_pm_supported(b, n, m, c, nsample, hidden, cout),
               "sa_mlp_pm: unsupported shape (C in {32,64,128}, nsample in {16,32,64}, npoint*nsample %% 128 == 0, hidden <= 128, out <= 256)");
    JM_REQUIRE(u_point_major && w1x && new_xyz && idx && w_hidden && b_hidden && w_out && b_out && out, "sa_mlp_pm: null pointer");
    JM_REQUIRE(((reinterpret_cast<uintptr_t>(u_point_major) | reinterpret_cast<uintptr_t>(w_hidden) | reinterpret_cast<uintptr_t>(w_out) |
                 reinterpret_cast<uintptr_t>(b_hidden)) & 15u) == 0, "sa_mlp_pm: 16-byte alignment");
    SaPmParams p{};
    p.N = n; p.M = m; p.C = c; p.ns = nsample;
    p.u = u_point_major; p.new_xyz = new_xyz; p.idx = idx; p.w1x = w1x;
    p.H = hidden; p.nblk1 = pad_to(hidden, 32) / 32; p.nkt1 = pad_to(hidden, 16) / 16;
    p.cout = cout; p.nblk2 = pad_to(cout, 32) / 32;
    p.np1 = pad_to(hidden, 128); p.np2 = pad_to(cout, 128);
    p.W1 = w_hidden; p.W2 = w_out; p.b1 = b_hidden; p.b2 = b_out; p.out = out;
    p.obs = obs ? obs : (size_t)cout * (size_t)m;
    p.S0 = c + 4; p.S1 = pad_to(hidden, 32) + 4;
    const size_t lds_bytes = sa_pm_lds_bytes(c, hidden);
    (void)hipFuncSetAttribute((const void*)sa_mlp_pm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8) cus = 256;
    cus -= cus % 8;
    p.tiles_per_frame = (int)((long long)m * nsample / PM_BM);
    p.total_tiles = (int)((long long)p.tiles_per_frame * b);
    p.xcd_frames = b >= 16 ? 1 : 0;
    p.total_dev = total_dev;
    if (total_dev) p.xcd_frames = 0;
#ifdef JM_TOOLS_BUILD
    p.dbg = tune_env("JM_PM_DBG", 0);
    p.trace = g_pm_trace;
    if (tune_env("JM_PM_LINEAR", 0)) p.xcd_frames = 0;
#endif
    const int grid = p.xcd_frames ? cus : (p.total_tiles < cus ? p.total_tiles : cus);
    hipLaunchKernelGGL(sa_mlp_pm_kernel, dim3((unsigned)grid), dim3(768), lds_bytes, (hipStream_t)stream, p);
    return check_launch("sa_mlp_pm");
}

extern "C" int jm_sa_mlp_pm_forward(int b, int n, int m, int c, int nsample, int hidden, int cout, const float* u_point_major,
                                    const float* w1x, const float* new_xyz, const int* idx, const float* w_hidden,
                                    const float* b_hidden, const float* w_out, const float* b_out, float* out,
                                    jm_stream_t stream) {
    return sa_mlp_pm_launch(b, n, m, c, nsample, hidden, cout, u_point_major, w1x, new_xyz, idx, w_hidden, b_hidden, w_out, b_out, out,
                            0, nullptr, stream);
}

/* the same with frame b's output at out + b * out_frame_stride floats (>= cout * m): a channel slice of a wider (B, Ctot, M) tensor */
extern "C" int jm_sa_mlp_pm_forward_into(int b, int n, int m, int c, int nsample, int hidden, int cout, const float* u_point_major,
                                         const float* w1x, const float* new_xyz, const int* idx, const float* w_hidden,
                                         const float* b_hidden, const float* w_out, const float* b_out, float* out,
                                         size_t out_frame_stride, jm_stream_t stream) {
    return sa_mlp_pm_launch(b, n, m, c, nsample, hidden, cout, u_point_major, w1x, new_xyz, idx, w_hidden, b_hidden, w_out, b_out, out,
                            out_frame_stride, nullptr, stream);
}

/* the same kernel on a row set whose size lives in device memory: frames = 1, m = the CAPACITY in (virtual) centres,
 * tiles_dev[0] = the number of 128-row tiles to run (<= m * nsample / 128); outputs of centres beyond it are not written */
extern "C" int jm_sa_mlp_pm_forward_dyn(int n, int m, int c, int nsample, int hidden, int cout, const float* u_point_major,
                                        const float* w1x, const float* new_xyz, const int* idx, const float* w_hidden,
                                        const float* b_hidden, const float* w_out, const float* b_out, float* out,
                                        const int* tiles_dev, jm_stream_t stream) {
    JM_REQUIRE(tiles_dev, "sa_mlp_pm_dyn: null tile count");
    return sa_mlp_pm_launch(1, n, m, c, nsample, hidden, cout, u_point_major, w1x, new_xyz, idx, w_hidden, b_hidden, w_out, b_out, out,
                            0, tiles_dev, stream);
}

/* LISTED form (csrc/sa_groups.hip): the same block on tiles of one class each, 2^q rows per group, outputs at the groups' own
 * positions; bit-identical to jm_sa_mlp_pm_forward_into.  jm_sa_mlp_pm_listed_qmin: the smallest class this shape tiles — 2 (row
 * quads), or 3 (octets) where the quads' table and partial maxima do not fit the LDS next to the tiles (C = hidden = 128), or -1
 * when the shape has no listed form; the plan must be made with that qmin (jm_sa_group_plan) */
extern "C" int jm_sa_mlp_pm_listed_qmin(int c, int hidden, int cout) {
    if ((c != 32 && c != 64 && c != 128) || hidden < 1 || hidden > 128 || cout < 1 || cout > PM_PW) return -1;
    for (int q = PM_QMIN_DEFAULT; q <= 3; ++q)
        if (sa_pm_listed_lds_bytes(c, hidden, cout, q) <= 160 * 1024) return q;
    return -1;
}

extern "C" int jm_sa_mlp_pm_listed_supported(int b, int n, int m, int c, int nsample, int hidden, int cout) {
    if (!jm_sa_mlp_pm_supported(b, n, m, c, nsample, hidden, cout)) return 0;
    if ((long long)b * m >= (1LL << 31) / 64) return 0;
    return jm_sa_mlp_pm_listed_qmin(c, hidden, cout) >= 0 ? 1 : 0;
}

extern "C" int jm_sa_mlp_pm_forward_listed(int b, int n, int m, int c, int nsample, int hidden, int cout, const float* u_point_major,
                                           const float* w1x, const float* new_xyz, const int* idx, const float* w_hidden,
                                           const float* b_hidden, const float* w_out, const float* b_out, const int* cls_count,
                                           const int* glist, float* out, size_t out_frame_stride, jm_stream_t stream) {
    JM_REQUIRE(b >= 0 && m >= 0, "sa_mlp_pm_listed: bad sizes");
    JM_REQUIRE(out_frame_stride == 0 || out_frame_stride >= (size_t)cout * (size_t)m, "sa_mlp_pm_listed: output frame stride below cout * npoint");
    if (b == 0 || m == 0) return JM_OK;
    JM_REQUIRE(jm_sa_mlp_pm_listed_supported(b, n, m, c, nsample, hidden, cout), "sa_mlp_pm_listed: unsupported shape");
    JM_REQUIRE(u_point_major && w1x && new_xyz && idx && w_hidden && b_hidden && w_out && b_out && out && cls_count && glist, "sa_mlp_pm_listed: null pointer");
    JM_REQUIRE(((reinterpret_cast<uintptr_t>(u_point_major) | reinterpret_cast<uintptr_t>(w_hidden) | reinterpret_cast<uintptr_t>(w_out) |
                 reinterpret_cast<uintptr_t>(b_hidden)) & 15u) == 0, "sa_mlp_pm_listed: 16-byte alignment");
    SaPmParams p{};
    p.N = n; p.M = m; p.C = c; p.ns = nsample;
    p.u = u_point_major; p.new_xyz = new_xyz; p.idx = idx; p.w1x = w1x;
    p.H = hidden; p.nblk1 = pad_to(hidden, 32) / 32; p.nkt1 = pad_to(hidden, 16) / 16;
    p.cout = cout; p.nblk2 = pad_to(cout, 32) / 32;
    p.np1 = pad_to(hidden, 128); p.np2 = pad_to(cout, 128);
    p.W1 = w_hidden; p.W2 = w_out; p.b1 = b_hidden; p.b2 = b_out; p.out = out;
    p.obs = out_frame_stride ? out_frame_stride : (size_t)cout * (size_t)m;
    p.S0 = c + 4; p.S1 = pad_to(hidden, 32) + 4;
    p.groups = b * m; p.qfull = nsample == 16 ? 4 : (nsample == 32 ? 5 : 6);
    p.qmin = jm_sa_mlp_pm_listed_qmin(c, hidden, cout);
    p.vt_floats = (int)sa_pm_listed_region(c, cout, p.qmin); p.p_cols = p.np2;
    p.cls_count = cls_count; p.glist = glist;
    const size_t lds_bytes = sa_pm_listed_lds_bytes(c, hidden, cout, p.qmin);
    (void)hipFuncSetAttribute((const void*)sa_mlp_pm_listed_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8) cus = 256;
    // tiles <= the dense count + one partial tile per class; persistent, one workgroup per CU at most
    const long long bound = (long long)b * m * nsample / PM_BM + (p.qfull - p.qmin + 1);
    const int grid = (int)(bound < cus ? bound : cus);
    hipLaunchKernelGGL(sa_mlp_pm_listed_kernel, dim3((unsigned)grid), dim3(768), lds_bytes, (hipStream_t)stream, p);
    return check_launch("sa_mlp_pm (listed)");
}
