// rcnn_lift.hip — the RCNN stage's per-point input MLP on the fp32 matrix cores (gfx950).
//
// Reference (jmodt/detection/modeling/rcnn.py:176-184): the pooled RoI points (R, S, 5 + C) = [xyz, mask, depth | C RPN
// feature channels] are split, the 5 geometric channels go through xyz_up_layer (SharedMLP 5 -> 128 -> 128), the RPN
// features are transposed, both are concatenated (R, 256, S, 1) and merge_down_layer (256 -> 128) produces the
// features the first set-abstraction level groups:  2 transposed copies + cat + three 1x1 convolutions, each a pass
// over a (R, 128..256, 512) tensor (268..537 MB at R = 1024).
//
// Here one launch: a tile = 32 consecutive pooled points = one contiguous 32 x (5 + C) block of the roipool3d output,
// staged k-major in LDS (a transposing, conflict-free copy), then
//     h1 = relu(W_u1 x5 + b);  h2 = relu(W_u2 h1 + b);  m = relu(W_m[:, :128] h2 + W_m[:, 128:] rpn + b)
// (the concatenation is two GEMMs accumulating in the same registers), optionally followed by the first
// set-abstraction layer hoisted in front of its gather (jm_sa_mlp_forward_pre):
//     u = W_1[:, 3:] m + W_1[:, :3] xyz + b_1                                   (no activation)
// and a (R, 128, S) store in the layout the SA kernels gather from.  Four waves split the columns of every stage;
// weights from L1/L2 in the packed layout of jm_sa_mlp_pack (jm_mfma.h: wide_ktiles).  Hidden widths <= 128.
#include "jm_mfma.h"

namespace jm {

constexpr int RL_XLD = 33;   // row stride of the transposed INPUT tiles: odd, so the transposing copy is conflict-free

struct RcnnLiftParams {
    int S, K, C;                       // points per RoI (multiple of 32), geometric channels (5), RPN feature channels
    int cp;                            // pad16(C)
    int h1, h2, hm, ho;                // layer widths (<= 128 each); ho = 0: no hoisted SA layer
    const float* pts;                  // (R, S, K + C)
    const float *Wu1, *Wu2, *WmH, *WmF, *WoM, *WoX;   // packed: (h1 x K), (h2 x h1), (hm x h2), (hm x C), (ho x hm), (ho x K)
    const float *bu1, *bu2, *bm, *bo;  // packed biases (128)
    float* out;                        // (R, hm or ho, S), or (R, S, hm or ho) with out_pm
    int out_pm;
    int tiles_per_roi;
    int total_tiles;                   // R * tiles_per_roi
    const int* work;                   // optional work list (see rcnn_lift_worklist_kernel): [0] = number of tiles, [1..] = tile ids
};

// tiles (roi * tiles_per_roi + tile) that hold at least one DISTINCT point of their slab: first point < max(count[roi], 1).
// One workgroup; work[0] = their number, work[1..] the ids in ascending order.
__global__ void __launch_bounds__(1024)
rcnn_lift_worklist_kernel(int R, int tiles_per_roi, const int* __restrict__ count, int* __restrict__ work) {
    __shared__ int wave_tot[16];
    __shared__ int base_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (int r0 = 0; r0 < R; r0 += 1024) {
        const int r = r0 + tid;
        const int nt = r < R ? min(tiles_per_roi, (max(count[r], 1) + SW_BM - 1) / SW_BM) : 0;
        const int incl = wave_incl_scan_i32_dpp(nt);
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        int off = base_s + incl - nt;
        for (int w = 0; w < wave; ++w) off += wave_tot[w];
        for (int t = 0; t < nt; ++t) work[1 + off + t] = r * tiles_per_roi + t;
        __syncthreads();
        if (tid == 1023) base_s = off + nt;
        __syncthreads();
    }
    if (tid == 0) work[0] = base_s;
}

__global__ void __launch_bounds__(256)
rcnn_lift_kernel(RcnnLiftParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lk = lane >> 5;
    const int a_off = lk * SW_LD + lr, x_off = lk * RL_XLD + lr;
    const int S = p.S, K = p.K, C = p.C, RC = K + C, cp = p.cp;
    float* X5 = lds;                                   // [16][33]  geometric channels (zero padded to 16)
    float* XF = X5 + 16 * RL_XLD;                      // [cp][33]  RPN features
    float* H1 = XF + (size_t)cp * RL_XLD;              // [128][36]
    float* H2 = H1 + 128 * SW_LD;                      // [128][36]
    // persistent: a workgroup walks tiles blockIdx.x, + gridDim.x, ... (16384 workgroups of 58 KB LDS cost ~40 ns each just
    // to be launched: 0.67 ms for this kernel's usual grid even when every one of them exits at once — measured)
    const int n_tiles = p.work ? min(p.work[0], p.total_tiles) : p.total_tiles;
    for (int wi = blockIdx.x; wi < n_tiles; wi += gridDim.x) {
    const int tile_id = p.work ? p.work[1 + wi] : wi;
    const int roi = tile_id / p.tiles_per_roi;
    const int s0 = (tile_id % p.tiles_per_roi) * SW_BM;
    // ---- transposing stage-in of the contiguous 32 x RC block: element e -> (row e / RC, channel e % RC)
    {
        const float* src = p.pts + ((size_t)roi * S + s0) * RC;
        const int total = SW_BM * RC;
        const unsigned magic = (unsigned)(0x100000000ULL / (unsigned)RC) + 1u;     // e / RC for e < 2^32 / RC
        for (int e = tid; e < total; e += 256) {
            const int row = (int)__umulhi((unsigned)e, magic), ch = e - row * RC;
            const float v = src[e];
            if (ch < K) X5[ch * RL_XLD + row] = v; else XF[(ch - K) * RL_XLD + row] = v;
        }
        for (int e = tid; e < (16 - K) * SW_BM; e += 256) X5[(K + e / SW_BM) * RL_XLD + (e % SW_BM)] = 0.f;
        for (int e = tid; e < (cp - C) * SW_BM; e += 256) XF[(C + e / SW_BM) * RL_XLD + (e % SW_BM)] = 0.f;
    }
    lds_barrier();

    const size_t woff = ((size_t)wave * 32 + lr) * 16 + lk * 8;     // one 32-column block per wave (widths <= 128)
    const size_t st = (size_t)128 * 16;
    auto set_bias = [=](f32x16& a, const float* bias) __attribute__((always_inline)) {
        const float bv = bias[wave * 32 + lr];
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] = bv;
    };
    auto store_hidden = [=](const f32x16& a, float* H, bool relu) __attribute__((always_inline)) {
        float* Hc = H + (size_t)(wave * 32 + lr) * SW_LD + 4 * lk;
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {                  // accumulator r = 4 rq + t  <->  row 8 rq + 4 lk + t
            float4 v;
            v.x = a[4 * rq + 0]; v.y = a[4 * rq + 1]; v.z = a[4 * rq + 2]; v.w = a[4 * rq + 3];
            if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            *reinterpret_cast<float4*>(Hc + 8 * rq) = v;
        }
    };
    f32x16 acc[2];
    // h1 = relu(W_u1 x5 + b)
    set_bias(acc[0], p.bu1);
    wide_ktiles<1>(X5, 1, p.Wu1 + woff, st, x_off, acc, RL_XLD);
    store_hidden(acc[0], H1, true);
    lds_barrier();
    // h2 = relu(W_u2 h1 + b)
    set_bias(acc[0], p.bu2);
    wide_ktiles<1>(H1, pad_to(p.h1, 16) / 16, p.Wu2 + woff, st, a_off, acc);
    store_hidden(acc[0], H2, true);
    lds_barrier();
    // m = relu(W_m[:, :h2] h2 + W_m[:, h2:] rpn + b)
    set_bias(acc[0], p.bm);
    wide_ktiles<1>(H2, pad_to(p.h2, 16) / 16, p.WmH + woff, st, a_off, acc);
    wide_ktiles<1>(XF, cp / 16, p.WmF + woff, st, x_off, acc, RL_XLD);
    int cout = p.hm;
    bool relu_out = true;
    if (p.ho > 0) {
        store_hidden(acc[0], H1, true);                   // H1 is free again: every wave passed the barrier after stage 2
        lds_barrier();
        // u = W_1[:, 3:] m + W_1[:, :3] xyz + b_1  (the hoisted first SA layer: linear, activation comes after the gather)
        set_bias(acc[0], p.bo);
        wide_ktiles<1>(H1, pad_to(p.hm, 16) / 16, p.WoM + woff, st, a_off, acc);
        wide_ktiles<1>(X5, 1, p.WoX + woff, st, x_off, acc, RL_XLD);
        cout = p.ho;
        relu_out = false;
    }
    const int col = wave * 32 + lr;
    if (col < cout && p.out_pm) {              // point-major rows: 32 lanes = 32 consecutive channels of one point
        float* o = p.out + ((size_t)roi * S + s0 + 4 * lk) * cout + col;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = acc[0][r];
            o[(size_t)((r & 3) + 8 * (r >> 2)) * cout] = relu_out ? fmaxf(v, 0.f) : v;
        }
    } else if (col < cout) {
        float* o = p.out + ((size_t)roi * cout + col) * S + s0 + 4 * lk;
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            float4 v;
            v.x = acc[0][4 * rq + 0]; v.y = acc[0][4 * rq + 1]; v.z = acc[0][4 * rq + 2]; v.w = acc[0][4 * rq + 3];
            if (relu_out) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            *reinterpret_cast<float4*>(o + 8 * rq) = v;
        }
    }
    lds_barrier();                                        // the tile's LDS operands are free for the next stage-in
    }
}

}  // namespace jm

using namespace jm;

extern "C" int jm_rcnn_lift_supported(int s, int k, int c, int h1, int h2, int hm, int ho) {
    if (s < 32 || s % 32 || k < 3 || k > 16 || c < 1 || h1 < 1 || h2 < 1 || hm < 1 || ho < 0) return 0;
    if (h1 > 128 || h2 > 128 || hm > 128 || ho > 128) return 0;
    if ((long long)32 * (k + c) * (k + c) >= (1LL << 32)) return 0;
    const size_t lds_bytes = ((size_t)(16 + pad_to(c, 16)) * RL_XLD + 2 * 128 * SW_LD) * sizeof(float);
    return lds_bytes <= 160 * 1024 ? 1 : 0;
}

/* all matrices in the layout of jm_sa_mlp_pack(cout, cin, first_layer = 0); w_out_m / w_out_x / b_out may be NULL (no
 * hoisted layer, h_out = 0): out = m (R, h_m, S); otherwise out = u (R, h_out, S) */
static int rcnn_lift_launch(int r, int s, int k, int c, int h1, int h2, int hm, int ho, const float* pts,
                            const float* w_up1, const float* b_up1, const float* w_up2, const float* b_up2,
                            const float* w_merge_h, const float* w_merge_f, const float* b_merge,
                            const float* w_out_m, const float* w_out_x, const float* b_out, int out_point_major,
                            float* out, const int* count, int* work, jm_stream_t stream) {
    JM_REQUIRE(r >= 0, "rcnn_lift: bad size");
    if (r == 0) return JM_OK;
    JM_REQUIRE(jm_rcnn_lift_supported(s, k, c, h1, h2, hm, ho), "rcnn_lift: unsupported shape (S %% 32 == 0, 3 <= K <= 16, widths <= 128)");
    JM_REQUIRE(pts && w_up1 && b_up1 && w_up2 && b_up2 && w_merge_h && w_merge_f && b_merge && out, "rcnn_lift: null pointer");
    JM_REQUIRE(ho == 0 || (w_out_m && w_out_x && b_out), "rcnn_lift: hoisted layer given without its weights");
    JM_REQUIRE((long long)r * (s / 32) < (1LL << 31), "rcnn_lift: too many tiles");
    JM_REQUIRE((reinterpret_cast<uintptr_t>(out) & 15u) == 0, "rcnn_lift: out must be 16-byte aligned");
    RcnnLiftParams p{};
    p.S = s; p.K = k; p.C = c; p.cp = pad_to(c, 16); p.h1 = h1; p.h2 = h2; p.hm = hm; p.ho = ho;
    p.pts = pts; p.Wu1 = w_up1; p.Wu2 = w_up2; p.WmH = w_merge_h; p.WmF = w_merge_f; p.WoM = w_out_m; p.WoX = w_out_x;
    p.bu1 = b_up1; p.bu2 = b_up2; p.bm = b_merge; p.bo = b_out; p.out = out; p.out_pm = out_point_major ? 1 : 0; p.tiles_per_roi = s / 32;
    p.total_tiles = (int)((long long)r * (s / 32));
    p.work = count ? work : nullptr;
    const size_t lds_bytes = ((size_t)(16 + p.cp) * RL_XLD + 2 * 128 * SW_LD) * sizeof(float);
    (void)hipFuncSetAttribute((const void*)rcnn_lift_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (count) hipLaunchKernelGGL(rcnn_lift_worklist_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, r, s / 32, count, work);
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
    // every tile live: one workgroup each (measured 787 us; a persistent grid of 8 per CU takes 823 us).  With a work list the
    // number of live tiles is unknown here: a persistent grid walks it (52 us at 11 distinct points per slab)
    const int grid = (count && p.total_tiles > 8 * cus) ? 8 * cus : p.total_tiles;
    hipLaunchKernelGGL(rcnn_lift_kernel, dim3((unsigned)grid), dim3(256), lds_bytes, (hipStream_t)stream, p);
    return check_launch("rcnn_lift");
}

extern "C" int jm_rcnn_lift_forward(int r, int s, int k, int c, int h1, int h2, int hm, int ho, const float* pts,
                                    const float* w_up1, const float* b_up1, const float* w_up2, const float* b_up2,
                                    const float* w_merge_h, const float* w_merge_f, const float* b_merge,
                                    const float* w_out_m, const float* w_out_x, const float* b_out, int out_point_major,
                                    float* out, jm_stream_t stream) {
    return rcnn_lift_launch(r, s, k, c, h1, h2, hm, ho, pts, w_up1, b_up1, w_up2, b_up2, w_merge_h, w_merge_f, b_merge, w_out_m, w_out_x,
                            b_out, out_point_major, out, nullptr, nullptr, stream);
}

/* the same on slabs whose rows count[r] .. S-1 are cyclic copies of rows 0 .. count[r]-1 (jm_roipool3d_canonical_cnt): 32-point
 * tiles that hold only copies are skipped, their output rows are NOT written — for consumers that read canonical rows only
 * (the duplicate-compacted set abstraction, csrc/sa_dedupe.hip).  work: (1 + r * s / 32) i32 scratch (the list of live tiles) */
extern "C" int jm_rcnn_lift_forward_cnt(int r, int s, int k, int c, int h1, int h2, int hm, int ho, const float* pts,
                                        const float* w_up1, const float* b_up1, const float* w_up2, const float* b_up2,
                                        const float* w_merge_h, const float* w_merge_f, const float* b_merge,
                                        const float* w_out_m, const float* w_out_x, const float* b_out, int out_point_major,
                                        float* out, const int* count, int* work, jm_stream_t stream) {
    JM_REQUIRE((count && work) || r == 0, "rcnn_lift: null count / work list");
    return rcnn_lift_launch(r, s, k, c, h1, h2, hm, ho, pts, w_up1, b_up1, w_up2, b_up2, w_merge_h, w_merge_f, b_merge, w_out_m, w_out_x,
                            b_out, out_point_major, out, count, work, stream);
}
