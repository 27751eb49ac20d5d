// conv1d_stack64.hip — the per-point Conv1d stacks of conv1d_stack.hip on 64-POINT tiles with the link head's operand recipe (gfx950).
//
// conv1d_stack.hip (round 2) works on 32-point tiles: four waves split the columns, the A operand is a k-major tile read with eight
// ds_read_b32 (or eight global loads, in place) per k-tile, and every weight register feeds ONE matrix instruction — on the stacks
// with the most rows (RPN heads rpn.py:34-58 and feature propagation level 1, pointnet2_modules.py:139-153: 131072 points each) the
// matrix pipe is busy 41 % of the cycles (profiles/r04_detect_pmc_MfmaUtil.txt).  The recipe that took the affinity link head from
// 0.76 to 0.83 of the fp32 peak (affinity_fused.hip), applied here:
//   * a workgroup owns 64 points from the inputs to the last layer; wave w owns 32-column blocks w, w + NW, ... of every layer for
//     all 64 rows: two 32 x 32 accumulator blocks, so every weight register feeds TWO v_mfma_f32_32x32x2_f32;
//   * the A operand of a layer is a ROW-major LDS tile T[point][K + 4]: the inputs (staged once: 64 consecutive points of a channel
//     are one 256-byte global row, transposed on the way in; two operands — or the point-major xyz operand of a hoisted
//     set-abstraction layer — side by side, the concatenation is never built), then the hidden activations written from the
//     accumulators (a lane half holds 32 consecutive columns of a row: conflict-free ds_write_b32);
//   * a lane's k-steps of a 16-deep k-tile are EIGHT CONSECUTIVE k (k = 16 kt + 8 (lane >> 5) + s): two ds_read_b128 per row block
//     and k-tile; the weights are packed once in B-operand order ([column block][k-tile][half][lane][4]: one coalesced 1 KB wave
//     load per four k-steps) and stream from L2 through a ring of four register sets, never through LDS;
//   * no barrier inside a layer; one after staging and one per hidden tile.
// The summation order inside a k-tile differs from conv1d_stack.hip's (2 s + (lane >> 5)): same products, 1e-4 parity with the fp32
// reference, not bit-identical to the 32-point kernel.  Shapes whose tiles do not fit the 160 KB LDS (feature propagation levels 2 /
// 3: 608 / 768 input channels) stay on the 32-point kernel.
#include <type_traits>

#include "jm_common.h"

namespace jm {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int C64_ROWS = 64, C64_RING = 4, C64_MAXL = 3;

struct CS64 {
    int n, tiles_per_frame;
    int c0, c1, xyz1;
    const float *x0, *x1;
    int L;
    int w[C64_MAXL];          // layer widths
    int nb[C64_MAXL];         // 32-column blocks per layer
    int kp[C64_MAXL];         // padded contraction length per layer (a multiple of 32)
    const float* W[C64_MAXL]; // packed weights
    const float* b[C64_MAXL]; // biases padded to nb * 32
    int relu[C64_MAXL];
    int off[C64_MAXL];        // LDS offset (floats) of layer l's INPUT tile
    float* out;
    int out_pm;
};

// W (n_out, k) row-major -> [column block][k-tile][h][lane = n % 32 + 32 ((k / 8) % 2)][k % 4], zero padded to nb * 32 columns and kp
// contraction elements; + the bias padded with zeros
__global__ void c64_pack_kernel(int n_out, int k, int kp, int nb, const float* __restrict__ W, int ldw, const float* __restrict__ bias,
                                float* __restrict__ dst, float* __restrict__ bdst) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)nb * 32 * kp;
    if (e < nb * 32) bdst[e] = (bias != nullptr && e < n_out) ? bias[e] : 0.f;
    if (e >= total) return;
    const int n = (int)(e / kp), kk_ = (int)(e - (long long)n * kp);
    const int cb = n >> 5, r = n & 31, kt = kk_ >> 4, kk = (kk_ >> 3) & 1, h = (kk_ >> 2) & 1, t = kk_ & 3;
    const float v = (n < n_out && kk_ < k) ? W[(size_t)n * ldw + kk_] : 0.f;
    dst[((((size_t)cb * (kp >> 4) + kt) * 2 + h) * 64 + (r + 32 * kk)) * 4 + t] = v;
}

__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 4)))
conv1d_stack64_kernel(CS64 p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NW = blockDim.x >> 6;
    const int lr = lane & 31, lk = lane >> 5;
    const int n = p.n;
    const int bi_ = blockIdx.x / p.tiles_per_frame;
    const int row0 = (blockIdx.x % p.tiles_per_frame) * C64_ROWS;

    // ---- stage the inputs: T0[point][channel], channels of operand 1 behind operand 0's, zero padding up to kp[0] ----
    {
        float* T0 = lds + p.off[0];
        const int ld = p.kp[0] + 4;
        const int pnt = lane;
        const float* x0 = p.x0 + (size_t)bi_ * p.c0 * n + row0 + pnt;
        const float* x1 = p.c1 == 0 ? nullptr
                        : (p.xyz1 ? p.x1 + ((size_t)bi_ * n + row0 + pnt) * 3 : p.x1 + (size_t)bi_ * p.c1 * n + row0 + pnt);
        const int ctot = p.c0 + p.c1;
        // eight channel rows in flight per wave (one load per iteration waits a full global-load latency per channel: 16 .. 64 of
        // them per wave and tile — measured, that WAS the kernel's time)
        for (int cbase = wave; cbase < p.kp[0]; cbase += 8 * NW) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int c = cbase + u * NW;
                float x = 0.f;
                if (c < p.c0) x = x0[(size_t)c * n];
                else if (c < ctot) x = p.xyz1 ? x1[c - p.c0] : x1[(size_t)(c - p.c0) * n];
                v[u] = x;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int c = cbase + u * NW;
                if (c < p.kp[0]) T0[(size_t)pnt * ld + c] = v[u];
            }
        }
    }
    __syncthreads();

    // one 32-column block of one layer over the 64-row LDS tile T (row stride ld), in HALF k-tiles (four k-steps = eight MFMAs): a
    // ring of C64_RING weight register sets runs C64_RING - 1 halves ahead of the MFMAs, the LDS operands one half ahead
    auto block = [&](auto two_c, const float* __restrict__ T, int ld, const float* __restrict__ wp, int KT, int cb, int rb, f32x16& acc0,
                     f32x16& acc1) __attribute__((always_inline)) {
        constexpr bool TWO = decltype(two_c)::value;           // both row blocks (acc0 / acc1), or row block rb only (acc0)
        const float4* bp = reinterpret_cast<const float4*>(wp) + (size_t)cb * KT * 128 + lane;
        const float* ap = T + (size_t)(lr + (TWO ? 0 : 32 * rb)) * ld + 8 * lk;
        const int NH = 2 * KT;                                 // (a multiple of 4: kp % 32 == 0)
        float4 b[C64_RING], a[2][2];
#define C64_LOAD_B(H) bp[(size_t)min((H), NH - 1) * 64]
#define C64_LOAD_A(H, BUF)                                                                       \
        {                                                                                        \
            const int hc = min((H), NH - 1);                                                     \
            a[BUF][0] = *reinterpret_cast<const float4*>(ap + 16 * (hc >> 1) + 4 * (hc & 1));    \
            if (TWO) a[BUF][1] = *reinterpret_cast<const float4*>(ap + (size_t)32 * ld + 16 * (hc >> 1) + 4 * (hc & 1)); \
        }
#define C64_STEP(BUF, E, BV)                                                                     \
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[BUF][0].E, BV, acc0, 0, 0, 0);         \
            if (TWO) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[BUF][1].E, BV, acc1, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < C64_RING - 1; ++j) b[j] = C64_LOAD_B(j);
        C64_LOAD_A(0, 0)
#define C64_PHASE(H0, J)                                                                          \
            {                                                                                    \
                b[((J) + C64_RING - 1) % C64_RING] = C64_LOAD_B((H0) + (J) + C64_RING - 1);      \
                C64_LOAD_A((H0) + (J) + 1, ((J) + 1) & 1)                                        \
                __builtin_amdgcn_sched_barrier(0);                                               \
                C64_STEP((J) & 1, x, b[(J) % C64_RING].x) C64_STEP((J) & 1, y, b[(J) % C64_RING].y) \
                C64_STEP((J) & 1, z, b[(J) % C64_RING].z) C64_STEP((J) & 1, w, b[(J) % C64_RING].w) \
                __builtin_amdgcn_sched_barrier(0);                                               \
            }
        for (int h0 = 0; h0 < NH; h0 += C64_RING) {
#pragma unroll
            for (int j = 0; j < C64_RING; ++j) C64_PHASE(h0, j)
        }
#undef C64_PHASE
#undef C64_LOAD_B
#undef C64_LOAD_A
#undef C64_STEP
    };

    for (int l = 0; l < p.L; ++l) {
        const float* T = lds + p.off[l];
        const int ld = p.kp[l] + 4;
        const int KT = p.kp[l] >> 4;
        const bool last = l + 1 == p.L;
        const float lo = p.relu[l] ? 0.f : -__builtin_inff();
        // work items: a layer with at least NW column blocks gives every wave whole 64-row blocks (two accumulators: each weight
        // register feeds two MFMAs); a narrower layer is split by ROW block as well, so that all waves have work
        const bool split = 2 * p.nb[l] <= NW || (p.nb[l] < NW && (p.nb[l] * 2) % NW == 0);
        const int items = split ? 2 * p.nb[l] : p.nb[l];
        for (int it = wave; it < items; it += NW) {
            const int cb = split ? it >> 1 : it, rb = split ? it & 1 : 0;
            const int col = cb * 32 + lr;
            const float bias = p.b[l][col];
            f32x16 acc0, acc1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[r] = bias; acc1[r] = bias; }
            if (split) block(std::false_type{}, T, ld, p.W[l], KT, cb, rb, acc0, acc1);
            else block(std::true_type{}, T, ld, p.W[l], KT, cb, 0, acc0, acc1);
            const int rsel = split ? 32 * rb : 0;             // first row of acc0's block
            // accumulator register r = row (r & 3) + 8 (r >> 2) + 4 lk of the block, column lr
            if (!last) {
                float* Tn = lds + p.off[l + 1];
                const int ldn = p.kp[l + 1] + 4;
                float* t = Tn + (size_t)(4 * lk + rsel) * ldn + col;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2);
                    t[(size_t)row * ldn] = fmaxf(acc0[r], lo);
                    if (!split) t[(size_t)(32 + row) * ldn] = fmaxf(acc1[r], lo);
                }
            } else if (col < p.w[l]) {
                const int oc = p.w[l];
                if (p.out_pm) {                                  // 32 lanes = 32 consecutive channels of one point
                    float* o = p.out + ((size_t)bi_ * n + row0 + 4 * lk + rsel) * oc + col;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (r & 3) + 8 * (r >> 2);
                        o[(size_t)row * oc] = fmaxf(acc0[r], lo);
                        if (!split) o[(size_t)(32 + row) * oc] = fmaxf(acc1[r], lo);
                    }
                } else {
                    float* o = p.out + ((size_t)bi_ * oc + col) * n + row0 + 4 * lk + rsel;
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {
                        float4 v0, v1;
                        v0.x = fmaxf(acc0[4 * rq + 0], lo); v0.y = fmaxf(acc0[4 * rq + 1], lo);
                        v0.z = fmaxf(acc0[4 * rq + 2], lo); v0.w = fmaxf(acc0[4 * rq + 3], lo);
                        v1.x = fmaxf(acc1[4 * rq + 0], lo); v1.y = fmaxf(acc1[4 * rq + 1], lo);
                        v1.z = fmaxf(acc1[4 * rq + 2], lo); v1.w = fmaxf(acc1[4 * rq + 3], lo);
                        *reinterpret_cast<float4*>(o + 8 * rq) = v0;
                        if (!split) *reinterpret_cast<float4*>(o + 32 + 8 * rq) = v1;
                    }
                }
            }
        }
        if (!last) __syncthreads();          // the hidden tile is complete (and every wave is done with this layer's input tile)
    }
}

static int pad32(int v) { return (v + 31) / 32 * 32; }

struct C64Plan { int kp[C64_MAXL], nb[C64_MAXL], off[C64_MAXL]; size_t lds_bytes; int waves; bool ok; };

// tiles: layer 0's input at offset 0, layer 1's behind it, layer 2's over layer 0's when it fits there (layer 0's tile is dead once
// every wave has passed the barrier behind layer 0), else behind layer 1's
static C64Plan c64_plan(int c0, int c1, int L, const int* w) {
    C64Plan pl{};
    int k = c0 + c1;
    size_t sz[C64_MAXL] = {0, 0, 0};
    int maxnb = 1;
    for (int l = 0; l < L; ++l) {
        pl.kp[l] = pad32(k);
        pl.nb[l] = (w[l] + 31) / 32;
        sz[l] = (size_t)C64_ROWS * (pl.kp[l] + 4);
        k = pl.nb[l] * 32;                       // the hidden tile holds the padded columns (zeros)
        if (pl.nb[l] > maxnb) maxnb = pl.nb[l];
    }
    pl.off[0] = 0;
    size_t total = sz[0];
    if (L >= 2) { pl.off[1] = (int)sz[0]; total = sz[0] + sz[1]; }
    if (L >= 3) {
        if (sz[2] <= sz[0]) pl.off[2] = 0;
        else { pl.off[2] = (int)(sz[0] + sz[1]); total += sz[2]; }
    }
    pl.lds_bytes = total * sizeof(float);
    pl.waves = maxnb >= 4 ? 8 : (maxnb >= 2 ? 4 : 2);          // (narrow layers are split by row block: conv1d_stack64_kernel)
    pl.ok = pl.lds_bytes <= 160 * 1024;
    return pl;
}

}  // namespace jm

using namespace jm;

extern "C" int jm_conv1d_stack64_supported(int b, int n, int c0, int c1, int xyz1, int num_layers, const int* widths) {
    if (b < 0 || n < 1 || c0 < 1 || c1 < 0 || num_layers < 1 || num_layers > C64_MAXL || !widths) return 0;
    if (xyz1 && c1 != 3) return 0;
    if (n % C64_ROWS || (long long)b * (n / C64_ROWS) >= (1LL << 31)) return 0;
    for (int l = 0; l < num_layers; ++l)
        if (widths[l] < 1) return 0;
    return c64_plan(c0, c1, num_layers, widths).ok ? 1 : 0;
}

extern "C" size_t jm_conv1d_stack64_packed_elems(int n_out, int k) { return (size_t)((n_out + 31) / 32 * 32) * (size_t)pad32(k); }

extern "C" int jm_conv1d_stack64_pack(int n_out, int k, const float* w, int ldw, const float* bias, float* packed, float* bias_padded,
                                      jm_stream_t stream) {
    JM_REQUIRE(n_out >= 1 && k >= 1 && w && ldw >= k && packed && bias_padded, "conv1d_stack64_pack: bad arguments");
    const int nb = (n_out + 31) / 32, kp = pad32(k);
    const long long total = (long long)nb * 32 * kp;
    hipLaunchKernelGGL(c64_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n_out, k, kp, nb, w, ldw, bias,
                       packed, bias_padded);
    return check_launch("conv1d_stack64_pack");
}

// layer l's packed weight covers its PADDED input width: pack layer 0 with k = c0 + c1, layer l > 0 with k = widths[l - 1] (the
// kernel's hidden tiles hold ceil(widths[l - 1] / 32) * 32 columns, the padding columns are zeros on both sides)
extern "C" int jm_conv1d_stack64_forward(int b, int n, int c0, const float* x0, int c1, const float* x1, int xyz1, int num_layers,
                                         const int* widths, const float* const* packed, const float* const* biases_padded, const int* relu,
                                         int out_point_major, float* out, jm_stream_t stream) {
    JM_REQUIRE(b >= 0 && n >= 0, "conv1d_stack64: bad sizes");
    if (b == 0 || n == 0) return JM_OK;
    JM_REQUIRE(jm_conv1d_stack64_supported(b, n, c0, c1, xyz1, num_layers, widths),
               "conv1d_stack64: unsupported shape (n %% 64 == 0, 1..3 layers, tiles within the 160 KB LDS)");
    JM_REQUIRE(x0 && packed && biases_padded && relu && out && (c1 == 0 || x1), "conv1d_stack64: null pointer");
    const C64Plan pl = c64_plan(c0, c1, num_layers, widths);
    CS64 p{};
    p.n = n; p.tiles_per_frame = n / C64_ROWS; p.c0 = c0; p.c1 = c1; p.xyz1 = xyz1; p.x0 = x0; p.x1 = x1; p.L = num_layers;
    for (int l = 0; l < num_layers; ++l) {
        JM_REQUIRE(packed[l] && biases_padded[l], "conv1d_stack64: null layer pointer");
        JM_REQUIRE((reinterpret_cast<uintptr_t>(packed[l]) & 15u) == 0, "conv1d_stack64: 16-byte alignment");
        p.w[l] = widths[l]; p.nb[l] = pl.nb[l]; p.kp[l] = pl.kp[l]; p.W[l] = packed[l]; p.b[l] = biases_padded[l]; p.relu[l] = relu[l];
        p.off[l] = pl.off[l];
    }
    JM_REQUIRE(out_point_major || (reinterpret_cast<uintptr_t>(out) & 15u) == 0, "conv1d_stack64: 16-byte alignment of the output");
    p.out = out; p.out_pm = out_point_major ? 1 : 0;
    (void)hipFuncSetAttribute((const void*)conv1d_stack64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(conv1d_stack64_kernel, dim3((unsigned)(b * (n / C64_ROWS))), dim3((unsigned)(64 * pl.waves)), pl.lds_bytes,
                       (hipStream_t)stream, p);
    return check_launch("conv1d_stack64");
}
