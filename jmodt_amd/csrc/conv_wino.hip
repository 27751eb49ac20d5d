// conv_wino.hip — 3x3 / stride 1 / padding 1 convolution + bias + ReLU on channels-last fp32 activations as ONE fused
// Winograd F(2x2, 3x3) kernel (gfx950).
//
// backbone.py:16-32: BasicBlock.conv1 + bn1 + relu of Img_Block[1..3] (64 -> 128 at 192 x 640, 128 -> 256 at 96 x 320,
// 256 -> 512 at 48 x 160; eval-mode BatchNorm folded into weight and bias by the caller): 3 x 145 GFLOP = 60 % of the image
// branch's arithmetic, which is half of the detect step.  As a direct convolution the fp32 MFMA pipe bounds each layer at
// 0.92 ms (157.3 TF); the library's implicit GEMM runs at 0.78-0.81 of that (1.14-1.19 ms) plus a bias/ReLU pass over the output.
// Winograd's minimal filtering needs 16 products per 2x2 output tile and (cin, cout) pair instead of 36: 64.4 GFLOP of
// MFMA work per layer, bound 0.41 ms.  Un-fused (transform kernels + 16 GEMMs + inverse transform) the transformed
// tensors move 6.7 GB per layer and lose to the direct form; here they never leave the CU:
//   * workgroup = 4 waves = 32 tiles (4 x 8: an 8 x 16 pixel output block, 10 x 18 input patch) x 64 output channels;
//     every wave owns ALL 16 transform positions of 32 tiles x 16 channels: 16 x 2 accumulator blocks of
//     v_mfma_f32_16x16x4_f32 (128 registers) — the inverse transform A^T M A is then lane-local arithmetic on the
//     accumulators, no exchange;
//   * the input transform B^T d B is done by the workgroup's 256 threads, one (tile, channel PAIR) each per 16-channel chunk:
//     16 buffer_load_dwordx2 (out-of-image taps read 0 through the buffer descriptor's range check — no masks), 32 packed
//     adds, 16 ds_write_b64 into a double-buffered V[16 positions][8 pairs][32 tiles][2] (32 KB per buffer), one barrier per
//     chunk.  The k order inside a chunk is permuted so that the two channels of a pair are the SAME lane's operand in two
//     consecutive MFMA k-steps (k-step 2j + e, lane quarter q <-> channel 2 (4 j + q) + e): one ds_read_b64 feeds two MFMAs.
//     tile' = tile + 16 (pair & 1) + 4 (pair >> 1) mod 32 keeps the 64-bit writes (16-lane groups) and reads (32-lane groups)
//     both free of bank conflicts;
//   * the transformed weights U = G g G^T are packed once, in MFMA B-operand order with the same k permutation ([16-channel
//     block][k-step][position group][lane][4]), and stream from L2 / L1 straight into registers as fully coalesced 1 KB wave
//     loads, three stages ahead: no LDS for the weights (the two workgroups resident on a CU share them through L1: the
//     default cache policy measured 10-20 % faster than non-temporal loads);
//   * PERSISTENT grid (two workgroups per CU, 246 registers): an XCD walks a contiguous range of (patch, 64-channel block)
//     items, channel-block-major inside groups of `group` patches, and the chunk pipeline runs ACROSS items: the windows and
//     the first weight stages of item n + 1 are in flight under the last chunks and the epilogue of item n; the window loads
//     are issued four per stage right behind the transform row that frees their registers.
// What bounds it (tools/wino_trace.py: shader-clock stamps per stage; tools/wino_pmc.sh): a wave needs ~8 k cycles per chunk
// for 4096 cycles of MFMA work — the vector-memory instructions cost their TA time wherever they are placed (16 weight + 16
// window loads per chunk and wave; TA busy 42-50 %) — and two waves per SIMD (the 128 accumulator registers) overlap that only
// partly: matrix pipe busy 58-69 %.  Halving the instruction count (8- vs 16-channel chunks), deeper operand prefetch and
// removing the per-item prologue each moved the time by 0-3 %; s_setprio around the MFMA groups (either way) costs 5 %;
// removing a feed (ablation) moves it by 10-25 %.
// Measured (tools/conv_wino_bench.py, batch 8): see DESIGN.md §4; error vs float64 3-6e-7 of the output range, the direct
// fp32 form 7e-7-1.4e-6.
#include <algorithm>

#include "jm_common.h"

namespace jm {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int WN_KC = 16;                    // input channels per chunk = four k-steps of the 16x16x4 MFMA
constexpr int WN_TN = 64;                    // output channels per workgroup (16 per wave)
constexpr int WN_ABUF = 1;                   // A-operand stage buffers (2 = fetched one stage ahead: +2 %, but 16 registers the spread window loads need)
constexpr int WN_POS = 8 * 32 * 2;           // floats per transform position in a V buffer: [pair][tile'][2]
constexpr int WN_VBUF = 16 * WN_POS;         // floats per V buffer (32 KB)

// channel (of a 16-channel chunk) that lane quarter kq multiplies in k-step ks of the chunk
__host__ __device__ constexpr int wn_channel(int ks, int kq) { return 2 * (4 * (ks >> 1) + kq) + (ks & 1); }
__device__ __forceinline__ int wn_slot(int pair, int tile) { return (pair * 32 + ((tile + 16 * (pair & 1) + 4 * (pair >> 1)) & 31)) * 2; }

// U[p][q] = sum_ab G[p][a] g[a][b] G[q][b], G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]; one thread per packed float4
__global__ void __launch_bounds__(256)
wino_pack_kernel(int cin, int cout, const float* __restrict__ w, float* __restrict__ up) {
    const int ksteps = cin >> 2;
    const long long total = (long long)(cout >> 4) * ksteps * 4 * 64;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int lane = (int)(i & 63), pg = (int)((i >> 6) & 3);
    const long long r = i >> 8;
    const int kstep = (int)(r % ksteps), nblk = (int)(r / ksteps);
    const int ci = WN_KC * (kstep >> 2) + wn_channel(kstep & 3, lane >> 4), co = 16 * nblk + (lane & 15);
    const float* g = w + ((size_t)co * cin + ci) * 9;
    double gg[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) gg[a][b] = (double)g[a * 3 + b];
    const double G[4][3] = {{1, 0, 0}, {.5, .5, .5}, {.5, -.5, .5}, {0, 0, 1}};
    float o[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        double s = 0;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) s += G[pg][a] * gg[a][b] * G[q][b];
        o[q] = (float)s;
    }
    reinterpret_cast<f32x4*>(up)[i] = (f32x4){o[0], o[1], o[2], o[3]};
}

#ifdef JM_TOOLS_BUILD
// tools/wino_trace.py: shader-clock stamps of every stage of sampled workgroups (LDS during the run, dumped at the end)
__device__ unsigned long long* g_wn_trace = nullptr;
constexpr int WN_TR = 11 * 16 + 4;           // stamps per wave: (top + 8 stages + loads + barrier) x up to 16 chunks + 4 marks
#define WN_STAMP(i) (tr[(i)] = __builtin_readcyclecounter())
#else
#define WN_STAMP(i) ((void)0)
#endif

__global__ void __launch_bounds__(256, 2)
conv3x3_wino_kernel(int H, int W, int cin, int cout, int patches_x, int patches_y, int npatches, int group, unsigned total_work, unsigned x_bytes,
                    const float* __restrict__ x, const float* __restrict__ up, const float* __restrict__ bias,
                    float* __restrict__ y, int relu) {
    __shared__ __attribute__((aligned(16))) float V[2 * WN_VBUF];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef JM_TOOLS_BUILD
    __shared__ unsigned long long trace[4 * WN_TR];
    unsigned long long* tr = trace + wave * WN_TR;
    WN_STAMP(WN_TR - 4);
#endif
    // ---- PERSISTENT walk: the grid is two workgroups per CU; a workgroup takes every (grid / 8)-th work item of its XCD's
    //      contiguous range.  Work item = (patch, 64-channel block), channel-block-major inside groups of `group` patches: the
    //      workgroups resident on an XCD stream the same few slices of U (L2-resident) instead of cycling through all of it. ----
    unsigned item, item_end, item_step;
    {
        const unsigned nwg = gridDim.x, bid = blockIdx.x;
        if ((total_work & 7u) == 0 && (nwg & 7u) == 0) {
            const unsigned n8 = total_work >> 3;
            item = (bid & 7u) * n8 + (bid >> 3); item_end = ((bid & 7u) + 1) * n8; item_step = nwg >> 3;
        } else { item = bid; item_end = total_work; item_step = nwg; }
    }
    if (item >= item_end) return;
    const int nblocks = cout / WN_TN;
    const unsigned per_group = (unsigned)group * (unsigned)nblocks;
    auto decode = [&](unsigned work, int& b, int& py, int& px, int& nb) __attribute__((always_inline)) {
        const unsigned grp = work / per_group, rem = work - grp * per_group;
        const unsigned gsize = min((unsigned)group, (unsigned)npatches - grp * (unsigned)group);
        nb = (int)(rem / gsize);
        int patch = (int)(grp * (unsigned)group + rem % gsize);
        px = patch % patches_x; patch /= patches_x;
        py = patch % patches_y;
        b = patch / patches_y;
    };

    // ---- input-transform role: one (tile, channel pair of the chunk) per thread.  Lane bits: [1:0] pair >> 1, [3:2] tile & 3,
    //      [4] pair & 1, [5] tile bit 2; wave = tile >> 3 (one row of 8 tiles): a 16-lane group writes 32 distinct banks ----
    const int tpair = 2 * (lane & 3) + ((lane >> 4) & 1), ttile = wave * 8 + ((lane >> 5) & 1) * 4 + ((lane >> 2) & 3);
    unsigned rrow[4], rcol[4];                 // byte offsets of the 4 rows / 4 columns of the tile's input window; 0xFFFFFF00 = outside
    auto window = [&](int b, int py, int px) __attribute__((always_inline)) {
        const int oy = (py * 4 + (ttile >> 3)) * 2 - 1, ox = (px * 8 + (ttile & 7)) * 2 - 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int iy = oy + i, ix = ox + i;
            rrow[i] = (iy >= 0 && iy < H) ? (unsigned)(((size_t)b * H + iy) * W) * (unsigned)cin * 4u + (unsigned)tpair * 8u : 0xFFFFFF00u;
            rcol[i] = (ix >= 0 && ix < W) ? (unsigned)ix * (unsigned)cin * 4u : 0xFFFFFF00u;
        }
    };
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (int)x_bytes, 0x00020000);
    const int wslot = wn_slot(tpair, ttile);

    // ---- MFMA role ----
    const int m = lane & 15, kq = lane >> 4;
    int aoff[2][2];                            // [j = k-step pair of the chunk][a = tile half]: float offset of the lane's 64-bit operand
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int a = 0; a < 2; ++a) aoff[j][a] = wn_slot(4 * j + kq, 16 * a + m);
    const int ksteps = cin >> 2, nch = cin / WN_KC, nst = nch * 8;
    auto wptr = [&](int nb) __attribute__((always_inline)) {
        return reinterpret_cast<const f32x4*>(up) + ((size_t)(nb * 4 + wave) * ksteps) * 256;   // wave-uniform: SGPR base, the lane part is the load's VGPR offset
    };

    f32x4 acc[16][2];
    f32x2 d[16];
    auto load_raw_row = [&](int i, int c) __attribute__((always_inline)) {      // window row i of chunk c: 4 x buffer_load_dwordx2
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // saturating sum: either part outside the image keeps the offset above the buffer's size
            const unsigned off = (rrow[i] | rcol[j]) >= 0xFFFFFF00u ? 0xFFFFFF00u : rrow[i] + rcol[j];
            d[i * 4 + j] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(xr, off, c * (WN_KC * 4), 0));
        }
    };
    auto load_raw = [&](int c) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) load_raw_row(i, c);
    };
    // B^T d B in two passes: columns (in place, 16 packed adds) and rows (16 packed adds + 16 LDS writes); the row pass is issued
    // one row per MFMA stage so that the vector work sits in the shadow of the matrix pipe
    auto transform_cols = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x2 t0 = d[0 + j] - d[8 + j], t1 = d[4 + j] + d[8 + j], t2 = d[8 + j] - d[4 + j], t3 = d[4 + j] - d[12 + j];
            d[0 + j] = t0; d[4 + j] = t1; d[8 + j] = t2; d[12 + j] = t3;
        }
    };
    auto transform_row = [&](float* Vb, int i) __attribute__((always_inline)) {
        f32x2* o = reinterpret_cast<f32x2*>(Vb + wslot);
        o[(i * 4 + 0) * (WN_POS / 2)] = d[i * 4 + 0] - d[i * 4 + 2];
        o[(i * 4 + 1) * (WN_POS / 2)] = d[i * 4 + 1] + d[i * 4 + 2];
        o[(i * 4 + 2) * (WN_POS / 2)] = d[i * 4 + 2] - d[i * 4 + 1];
        o[(i * 4 + 3) * (WN_POS / 2)] = d[i * 4 + 1] - d[i * 4 + 3];
    };
    // one stage = 4 positions x 2 k-steps = 16 MFMAs: its A operands are 8 ds_read_b64 (one stage ahead), its B operands two 1 KB
    // wave loads (ring of 4: three stages ahead)
    f32x4 bq[4][2];
    f32x2 af[WN_ABUF][8];
    auto load_b = [&](int slot, const f32x4* base, int stage) __attribute__((always_inline)) {   // stage = 8 * chunk + 4 * j + g of ITS item
        const size_t ks0 = (size_t)(stage >> 2) * 2;                          // first k-step of the pair
        bq[slot][0] = base[ks0 * 256 + (stage & 3) * 64 + lane];
        bq[slot][1] = base[(ks0 + 1) * 256 + (stage & 3) * 64 + lane];
    };
    auto load_a = [&](const float* Vb, int stage) __attribute__((always_inline)) {
        const int j = stage >> 2, g = stage & 3;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            af[stage % WN_ABUF][2 * q] = *reinterpret_cast<const f32x2*>(Vb + (g * 4 + q) * WN_POS + aoff[j][0]);
            af[stage % WN_ABUF][2 * q + 1] = *reinterpret_cast<const f32x2*>(Vb + (g * 4 + q) * WN_POS + aoff[j][1]);
        }
    };
    auto mma = [&](int stage) __attribute__((always_inline)) {
        const int g = stage & 3;
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float bv = bq[stage & 3][e][q];
                acc[g * 4 + q][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[stage % WN_ABUF][2 * q][e], bv, acc[g * 4 + q][0], 0, 0, 0);
                acc[g * 4 + q][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[stage % WN_ABUF][2 * q + 1][e], bv, acc[g * 4 + q][1], 0, 0, 0);
            }
    };

    // ---- the chunk pipeline runs ACROSS work items: while item n's last chunks are multiplied, the windows of item n + 1's first
    //      two chunks are requested / transformed and its first weight stages are in flight, so only the very first item of a
    //      workgroup has a prologue; between items there is just the inverse transform + store of the finished accumulators.
    //      Cursors: (cb, cpy, cpx, cnb) = the item being multiplied; (ld_item, ld_c) = the chunk whose window is requested next;
    //      ub / ub_next = the weight slices of this / the next item.  All loop bodies are branch-free around the loads (indices
    //      and pointers selected, not branched on): with conditional loads hipcc's s_waitcnt placement falls back to vmcnt(0..1) ----
    int cb, cpy, cpx, cnb;
    decode(item, cb, cpy, cpx, cnb);
    window(cb, cpy, cpx);
    unsigned ld_item = item;
    int ld_c = 0;
    auto advance_window = [&]() __attribute__((always_inline)) {   // ALU only inside the branch: the pending-load state is the same on both paths
        if (++ld_c == nch) {
            ld_c = 0;
            if (ld_item + item_step < item_end) {
                ld_item += item_step;
                int b2, py2, px2, nb2;
                decode(ld_item, b2, py2, px2, nb2);
                window(b2, py2, px2);
            }                                                      // past the last item: the same windows again (never used)
        }
    };
    const f32x4* ub = wptr(cnb);
    const f32x4* ub_next = ub;
    auto next_weights = [&]() __attribute__((always_inline)) {
        if (item + item_step < item_end) {
            int b2, py2, px2, nb2;
            decode(item + item_step, b2, py2, px2, nb2);
            ub_next = wptr(nb2);
        }
    };
    next_weights();
    float sb[16];
    auto load_bias = [&]() __attribute__((always_inline)) {
        const float* bp = bias + cnb * WN_TN + wave * 16;          // wave-uniform: s_load_dwordx16
#pragma unroll
        for (int k = 0; k < 16; ++k) sb[k] = bias ? bp[k] : 0.f;
    };
    load_bias();
    load_raw(ld_c); advance_window();
    load_b(0, ub, 0); load_b(1, ub, 1); load_b(2, ub, 2);
    transform_cols();
#pragma unroll
    for (int i = 0; i < 4; ++i) transform_row(V, i);
    load_raw(ld_c); advance_window();
    __syncthreads();
    int par = 0;
    for (;;) {
#pragma unroll
        for (int p = 0; p < 16; ++p) acc[p][0] = acc[p][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < nch; ++c) {
            const float* Vc = V + par * WN_VBUF;
            float* Vn = V + (par ^ 1) * WN_VBUF;
            if (WN_ABUF == 2) load_a(Vc, 0);
            WN_STAMP((c & 15) * 11);
#pragma unroll
            for (int st = 0; st < 8; ++st) {
                if (WN_ABUF == 1) load_a(Vc, st);
                else if (st < 7) load_a(Vc, st + 1);
                {
                    const int s = c * 8 + st + 3;
                    const bool over = s >= nst;                   // the next item's first stages
                    load_b((st + 3) & 3, over ? ub_next : ub, over ? s - nst : s);
                }
                mma(st);
                if (st == 3) transform_cols();   // the next chunk's window was requested during stages 4-7 of the previous iteration
                if (st >= 4) {
                    transform_row(Vn, st - 4);   // frees d[4 i .. 4 i + 3] ...
                    load_raw_row(st - 4, ld_c);  // ... for row i of the window after next: 4 loads per stage, under this stage's MFMAs
                }                                // (all 16 at the end of the chunk cost 1250 cycles of issue in front of the barrier)
                WN_STAMP((c & 15) * 11 + 1 + st);
                __builtin_amdgcn_sched_barrier(0);
            }
            advance_window();
            WN_STAMP((c & 15) * 11 + 9);
            __syncthreads();
            WN_STAMP((c & 15) * 11 + 10);
            par ^= 1;
        }

        // ---- inverse transform Y = A^T M A (A^T = [[1,1,1,0],[0,1,-1,-1]]), bias, ReLU, store ----
        WN_STAMP(WN_TR - 3);
        const int co = cnb * WN_TN + wave * 16 + m;
        float bv = 0.f;                         // the wave's 16 biases were fetched through the scalar cache when the item started: a vector
#pragma unroll                                  // load here would wait (in-order vmcnt) for every window / weight load in flight for the next item
        for (int k = 0; k < 16; ++k) bv = (m == k) ? sb[k] : bv;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int tile = 16 * a + 4 * kq + i;
                const int oy = (cpy * 4 + (tile >> 3)) * 2, ox = (cpx * 8 + (tile & 7)) * 2;
                float r0[4], r1[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    r0[q] = acc[0 + q][a][i] + acc[4 + q][a][i] + acc[8 + q][a][i];
                    r1[q] = acc[4 + q][a][i] - acc[8 + q][a][i] - acc[12 + q][a][i];
                }
                float o00 = r0[0] + r0[1] + r0[2] + bv, o01 = r0[1] - r0[2] - r0[3] + bv;
                float o10 = r1[0] + r1[1] + r1[2] + bv, o11 = r1[1] - r1[2] - r1[3] + bv;
                if (relu) { o00 = fmaxf(o00, 0.f); o01 = fmaxf(o01, 0.f); o10 = fmaxf(o10, 0.f); o11 = fmaxf(o11, 0.f); }
                if (oy < H && ox < W) {
                    float* o = y + (((size_t)cb * H + oy) * W + ox) * cout + co;
                    o[0] = o00;
                    if (ox + 1 < W) o[cout] = o01;
                    if (oy + 1 < H) {
                        o[(size_t)W * cout] = o10;
                        if (ox + 1 < W) o[(size_t)W * cout + cout] = o11;   // (non-temporal stores measured 1-3 % slower here)
                    }
                }
            }
        WN_STAMP(WN_TR - 2);
        item += item_step;
        if (item >= item_end) break;
        decode(item, cb, cpy, cpx, cnb);
        ub = ub_next;
        next_weights();
        load_bias();
    }
#ifdef JM_TOOLS_BUILD
    WN_STAMP(WN_TR - 1);
    if (g_wn_trace && blockIdx.x % 61 == 10) {            // eight workgroups spread over the grid (their LAST item's stamps)
        __syncthreads();
        unsigned long long* dst = g_wn_trace + (size_t)(blockIdx.x / 61) * (4 * WN_TR + 4);
        for (int i = tid; i < 4 * WN_TR; i += 256) dst[i] = trace[i];
        if (tid == 0) { dst[4 * WN_TR] = blockIdx.x; dst[4 * WN_TR + 1] = nch; dst[4 * WN_TR + 2] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)); }
    }
#endif
}

}  // namespace jm

using namespace jm;

#ifdef JM_TOOLS_BUILD
extern "C" __attribute__((visibility("default"))) int jm_tools_wino_trace(unsigned long long* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_wn_trace), &buf, sizeof(buf)) == hipSuccess ? 0 : 1;
}
extern "C" __attribute__((visibility("default"))) int jm_tools_wino_trace_stride(void) { return 4 * WN_TR + 4; }
#endif

extern "C" size_t jm_conv3x3_wino_packed_elems(int cin, int cout) { return (size_t)16 * (size_t)cin * (size_t)cout; }

extern "C" int jm_conv3x3_wino_supported(int cin, int cout) { return cin >= WN_KC && cin % WN_KC == 0 && cout >= WN_TN && cout % WN_TN == 0; }

extern "C" int jm_conv3x3_wino_pack(int cin, int cout, const float* weight, float* packed, jm_stream_t stream) {
    JM_REQUIRE(jm_conv3x3_wino_supported(cin, cout), "conv3x3_wino_pack: cin %% 16 == 0 and cout %% 64 == 0");
    JM_REQUIRE(weight && packed, "conv3x3_wino_pack: null pointer");
    JM_REQUIRE((reinterpret_cast<uintptr_t>(packed) & 15u) == 0, "conv3x3_wino_pack: 16-byte alignment");
    const long long total = (long long)(cout >> 4) * (cin >> 2) * 256;
    hipLaunchKernelGGL(wino_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, cin, cout, weight, packed);
    return check_launch("conv3x3_wino_pack");
}

extern "C" int jm_conv3x3_wino_bias_relu(int b, int h, int w, int cin, int cout, const float* x_channels_last,
                                         const float* packed, const float* bias, int relu, float* out_channels_last,
                                         jm_stream_t stream) {
    JM_REQUIRE(b >= 0 && h >= 0 && w >= 0, "conv3x3_wino: negative size");
    JM_REQUIRE(jm_conv3x3_wino_supported(cin, cout), "conv3x3_wino: cin %% 16 == 0 and cout %% 64 == 0");
    if (b == 0 || h == 0 || w == 0) return JM_OK;
    JM_REQUIRE(x_channels_last && packed && out_channels_last, "conv3x3_wino: null pointer");
    JM_REQUIRE(((reinterpret_cast<uintptr_t>(packed) | reinterpret_cast<uintptr_t>(x_channels_last)) & 15u) == 0, "conv3x3_wino: 16-byte alignment");
    const unsigned long long xb = (unsigned long long)b * h * w * cin * 4ull;
    JM_REQUIRE(xb < 0xFFFFFF00ull, "conv3x3_wino: input above 4 GB");
    const int tx = divup(w, 2), ty = divup(h, 2), pxs = divup(tx, 8), pys = divup(ty, 4);
    const unsigned long long total = (unsigned long long)b * pxs * pys * (cout / WN_TN);
    JM_REQUIRE(total < 0x7FFFFFFFull, "conv3x3_wino: grid limit");
    const int npatches = b * pxs * pys, group = tune_env("JM_WN_G", 16);
    // (tools build: JM_WN_GRID < 512 leaves CUs to the kernels of other streams — the persistent workgroups fill every SIMD's register file)
    const unsigned grid = (unsigned)std::min<unsigned long long>(total, (unsigned long long)tune_env("JM_WN_GRID", 2 * 256 * tune_env("JM_WN_WGS", 1)));
    hipLaunchKernelGGL(conv3x3_wino_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, h, w, cin, cout, pxs, pys, npatches, group,
                       (unsigned)total, (unsigned)xb, x_channels_last, packed, bias, out_channels_last, relu);
    return check_launch("conv3x3_wino");
}
