// feature_gather.hip — LI-Fusion point -> image bilinear gather for gfx950.
//
// Replaces feature_gather (jmodt/detection/modeling/backbone.py:79-89) =
//   F.grid_sample(feature_map, xy[B,1,N,2], mode='bilinear', padding_mode='zeros',
//                 align_corners=True).squeeze(2)
//
// Design: lane = point.  The 4 tap offsets / weights / validity are computed once per point and
// reused for a block of channels; the feature map is addressed through element strides, so a
// channels-last map (each tap = C contiguous floats) needs no copy and turns the 4*C scattered
// 4-byte reads of the NCHW layout into 16-byte vector reads.  Output (B,C,N) stores are
// contiguous along N.  Arithmetic follows ATen's grid_sampler_2d (unnormalise with
// align_corners, corner weights as products of differences, out-of-range taps contribute 0).
#include "jm_common.h"

namespace jm {

struct Taps {
    long long o_nw, o_ne, o_sw, o_se;  // element offsets inside one (b, c=0) plane walk
    float w_nw, w_ne, w_sw, w_se;      // 0 where the tap is outside the image
};

__device__ __forceinline__ Taps make_taps(float x, float y, int H, int W, long long sh, long long sw) {
    Taps t;
    const float ix = ((x + 1.f) / 2) * (W - 1);
    const float iy = ((y + 1.f) / 2) * (H - 1);
    const float fx = floorf(ix), fy = floorf(iy);
    const float nw = (fx + 1 - ix) * (fy + 1 - iy);
    const float ne = (ix - fx) * (fy + 1 - iy);
    const float sw_ = (fx + 1 - ix) * (iy - fy);
    const float se = (ix - fx) * (iy - fy);
    // compare in float before converting so huge / NaN coordinates cannot overflow the int cast
    const bool x0ok = fx >= 0.f && fx <= (float)(W - 1), x1ok = fx + 1 >= 0.f && fx + 1 <= (float)(W - 1);
    const bool y0ok = fy >= 0.f && fy <= (float)(H - 1), y1ok = fy + 1 >= 0.f && fy + 1 <= (float)(H - 1);
    const int x0 = x0ok ? (int)fx : 0, x1 = x1ok ? (int)fx + 1 : 0;
    const int y0 = y0ok ? (int)fy : 0, y1 = y1ok ? (int)fy + 1 : 0;
    t.o_nw = y0 * sh + x0 * sw; t.o_ne = y0 * sh + x1 * sw;
    t.o_sw = y1 * sh + x0 * sw; t.o_se = y1 * sh + x1 * sw;
    t.w_nw = (x0ok && y0ok) ? nw : 0.f;
    t.w_ne = (x1ok && y0ok) ? ne : 0.f;
    t.w_sw = (x0ok && y1ok) ? sw_ : 0.f;
    t.w_se = (x1ok && y1ok) ? se : 0.f;
    return t;
}

constexpr int FG_CPT = 16;

// generic strides (NCHW and anything else)
__global__ void __launch_bounds__(256)
feature_gather_kernel(int C, int H, int W, int N, const float* __restrict__ fmap, long long sb, long long sc,
                      long long sh, long long sw, const float* __restrict__ xy, float* __restrict__ out) {
    const int bi = blockIdx.z;
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float2 p = *reinterpret_cast<const float2*>(xy + ((size_t)bi * N + n) * 2);
    const Taps t = make_taps(p.x, p.y, H, W, sh, sw);
    const int c0 = blockIdx.y * FG_CPT, c1 = min(C, c0 + FG_CPT);
    const float* base = fmap + (size_t)bi * sb;
#pragma unroll 4
    for (int c = c0; c < c1; ++c) {
        const float* pl = base + (size_t)c * sc;
        // taps with zero weight are still read from a clamped in-range address (weight 0 kills them)
        float acc = pl[t.o_nw] * t.w_nw;
        acc += pl[t.o_ne] * t.w_ne;
        acc += pl[t.o_sw] * t.w_sw;
        acc += pl[t.o_se] * t.w_se;
        out[((size_t)bi * C + c) * N + n] = acc;
    }
}

// NCHW-like maps (unit stride along x, W >= 2): the two taps of a row are neighbours in memory, so ONE
// 8-byte load per row replaces two 4-byte gathers — half the sectors pulled through the fabric and half
// the vector-memory instructions of the generic kernel.  The pair starts at xb = clamp(x0, 0, W-2);
// which element belongs to which tap is decided once per point.  Same products, same summation order.
struct __attribute__((packed, aligned(4))) F2u { float x, y; };

__global__ void __launch_bounds__(256)
feature_gather_rowpair_kernel(int C, int H, int W, int N, const float* __restrict__ fmap, long long sb, long long sc,
                              long long sh, const float* __restrict__ xy, float* __restrict__ out) {
    const int bi = blockIdx.z;
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float2 p = *reinterpret_cast<const float2*>(xy + ((size_t)bi * N + n) * 2);
    const Taps t = make_taps(p.x, p.y, H, W, sh, 1);
    // recover the (clamped) tap columns from the offsets make_taps produced: o = y * sh + x
    const long long row0 = (t.o_nw / sh) * sh, row1 = (t.o_sw / sh) * sh;
    const int x0 = (int)(t.o_nw - row0), x1 = (int)(t.o_ne - row0);
    const int xb = min(max(x0, 0), W - 2);
    const bool w0_first = (x0 == xb), e0_first = (x1 == xb);   // which element of the pair each tap reads
    const long long o0 = row0 + xb, o1 = row1 + xb;
    const int c0 = blockIdx.y * FG_CPT, c1 = min(C, c0 + FG_CPT);
    const float* base = fmap + (size_t)bi * sb;
#pragma unroll 4
    for (int c = c0; c < c1; ++c) {
        const float* pl = base + (size_t)c * sc;
        const F2u r0 = *reinterpret_cast<const F2u*>(pl + o0);
        const F2u r1 = *reinterpret_cast<const F2u*>(pl + o1);
        float acc = (w0_first ? r0.x : r0.y) * t.w_nw;
        acc += (e0_first ? r0.x : r0.y) * t.w_ne;
        acc += (w0_first ? r1.x : r1.y) * t.w_sw;
        acc += (e0_first ? r1.x : r1.y) * t.w_se;
        out[((size_t)bi * C + c) * N + n] = acc;
    }
}

// channels-last fast path: sc == 1, C % 4 == 0, 16-byte aligned taps
__global__ void __launch_bounds__(256)
feature_gather_cl_kernel(int C, int H, int W, int N, const float* __restrict__ fmap, long long sb, long long sh,
                         long long sw, const float* __restrict__ xy, float* __restrict__ out) {
    const int bi = blockIdx.z;
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float2 p = *reinterpret_cast<const float2*>(xy + ((size_t)bi * N + n) * 2);
    const Taps t = make_taps(p.x, p.y, H, W, sh, sw);
    const int c0 = blockIdx.y * FG_CPT, c1 = min(C, c0 + FG_CPT);
    const float* base = fmap + (size_t)bi * sb;
    for (int c = c0; c < c1; c += 4) {
        const float4 a = *reinterpret_cast<const float4*>(base + t.o_nw + c);
        const float4 b = *reinterpret_cast<const float4*>(base + t.o_ne + c);
        const float4 d = *reinterpret_cast<const float4*>(base + t.o_sw + c);
        const float4 e = *reinterpret_cast<const float4*>(base + t.o_se + c);
        float r[4];
        r[0] = a.x * t.w_nw; r[0] += b.x * t.w_ne; r[0] += d.x * t.w_sw; r[0] += e.x * t.w_se;
        r[1] = a.y * t.w_nw; r[1] += b.y * t.w_ne; r[1] += d.y * t.w_sw; r[1] += e.y * t.w_se;
        r[2] = a.z * t.w_nw; r[2] += b.z * t.w_ne; r[2] += d.z * t.w_sw; r[2] += e.z * t.w_se;
        r[3] = a.w * t.w_nw; r[3] += b.w * t.w_ne; r[3] += d.w * t.w_sw; r[3] += e.w * t.w_se;
#pragma unroll
        for (int q = 0; q < 4; ++q) out[((size_t)bi * C + c + q) * N + n] = r[q];
    }
}

__global__ void __launch_bounds__(256)
feature_gather_grad_kernel(int C, int H, int W, int N, const float* __restrict__ grad_out,
                           const float* __restrict__ xy, float* __restrict__ grad_fmap, long long sb, long long sc,
                           long long sh, long long sw) {
    const int bi = blockIdx.z;
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float2 p = *reinterpret_cast<const float2*>(xy + ((size_t)bi * N + n) * 2);
    const Taps t = make_taps(p.x, p.y, H, W, sh, sw);
    const int c0 = blockIdx.y * FG_CPT, c1 = min(C, c0 + FG_CPT);
    float* base = grad_fmap + (size_t)bi * sb;
    for (int c = c0; c < c1; ++c) {
        const float g = grad_out[((size_t)bi * C + c) * N + n];
        float* pl = base + (size_t)c * sc;
        if (t.w_nw != 0.f) unsafeAtomicAdd(pl + t.o_nw, g * t.w_nw);
        if (t.w_ne != 0.f) unsafeAtomicAdd(pl + t.o_ne, g * t.w_ne);
        if (t.w_sw != 0.f) unsafeAtomicAdd(pl + t.o_sw, g * t.w_sw);
        if (t.w_se != 0.f) unsafeAtomicAdd(pl + t.o_se, g * t.w_se);
    }
}

// ---- training path: ROW-major output (B N, C) from a channels-last map; thread = (point, 4 channels), consecutive lanes =
// consecutive channel quads of one point: the four tap vectors are contiguous 16-byte reads, the output row one contiguous store
__global__ void __launch_bounds__(256)
feature_gather_rows_kernel(int B, int C, int H, int W, int N, const float* __restrict__ fmap, const float* __restrict__ xy,
                           float* __restrict__ out, int ldo) {
    const int q = C / 4;
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= (long long)B * N * q) return;
    const int pt = (int)(i / q), c = (int)(i % q) * 4;
    const int bi = pt / N;
    const float2 p = *reinterpret_cast<const float2*>(xy + (size_t)pt * 2);
    const Taps t = make_taps(p.x, p.y, H, W, (long long)W * C, C);
    const float* base = fmap + (size_t)bi * H * W * C + c;
    const float4 a = *reinterpret_cast<const float4*>(base + t.o_nw);
    const float4 b = *reinterpret_cast<const float4*>(base + t.o_ne);
    const float4 d = *reinterpret_cast<const float4*>(base + t.o_sw);
    const float4 e = *reinterpret_cast<const float4*>(base + t.o_se);
    float r[4];
    r[0] = a.x * t.w_nw; r[0] += b.x * t.w_ne; r[0] += d.x * t.w_sw; r[0] += e.x * t.w_se;
    r[1] = a.y * t.w_nw; r[1] += b.y * t.w_ne; r[1] += d.y * t.w_sw; r[1] += e.y * t.w_se;
    r[2] = a.z * t.w_nw; r[2] += b.z * t.w_ne; r[2] += d.z * t.w_sw; r[2] += e.z * t.w_se;
    r[3] = a.w * t.w_nw; r[3] += b.w * t.w_ne; r[3] += d.w * t.w_sw; r[3] += e.w * t.w_se;
    *reinterpret_cast<float4*>(out + (size_t)pt * ldo + c) = make_float4(r[0], r[1], r[2], r[3]);
}

// thread = (point, channel): a wave adds 64 consecutive channels of one tap = 256 contiguous bytes per atomic instruction
__global__ void __launch_bounds__(256)
feature_gather_rows_grad_kernel(int B, int C, int H, int W, int N, const float* __restrict__ grad_out, int ldo, const float* __restrict__ xy,
                                float* __restrict__ grad_fmap) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= (long long)B * N * C) return;
    const int pt = (int)(i / C), c = (int)(i % C);
    const int bi = pt / N;
    const float2 p = *reinterpret_cast<const float2*>(xy + (size_t)pt * 2);
    const Taps t = make_taps(p.x, p.y, H, W, (long long)W * C, C);
    const float g = grad_out[(size_t)pt * ldo + c];
    float* base = grad_fmap + (size_t)bi * H * W * C + c;
    if (t.w_nw != 0.f) unsafeAtomicAdd(base + t.o_nw, g * t.w_nw);
    if (t.w_ne != 0.f) unsafeAtomicAdd(base + t.o_ne, g * t.w_ne);
    if (t.w_sw != 0.f) unsafeAtomicAdd(base + t.o_sw, g * t.w_sw);
    if (t.w_se != 0.f) unsafeAtomicAdd(base + t.o_se, g * t.w_se);
}

}  // namespace jm

using namespace jm;

extern "C" int jm_feature_gather(int b, int c, int h, int w, int n, const float* fmap, int64_t sb, int64_t sc,
                                 int64_t sh, int64_t sw, const float* xy, float* out, jm_stream_t stream) {
    JM_REQUIRE(b >= 0 && c >= 0 && h >= 1 && w >= 1 && n >= 0, "feature_gather: bad sizes");
    if (b == 0 || c == 0 || n == 0) return JM_OK;
    JM_REQUIRE(fmap && xy && out, "feature_gather: null pointer");
    JM_REQUIRE(b <= 65535 && divup(c, FG_CPT) <= 65535, "feature_gather: shape too large");
    JM_REQUIRE((reinterpret_cast<uintptr_t>(xy) & 7u) == 0, "feature_gather: xy must be 8-byte aligned");
    dim3 grid(divup(n, 256), divup(c, FG_CPT), b), block(256);
    const bool cl = sc == 1 && (c % 4 == 0) && (sw % 4 == 0) && (sh % 4 == 0) && (sb % 4 == 0) &&
                    ((reinterpret_cast<uintptr_t>(fmap) & 15u) == 0);
    if (cl)
        hipLaunchKernelGGL(feature_gather_cl_kernel, grid, block, 0, (hipStream_t)stream, c, h, w, n, fmap,
                           (long long)sb, (long long)sh, (long long)sw, xy, out);
    else if (sw == 1 && w >= 2 && sh >= w)
        hipLaunchKernelGGL(feature_gather_rowpair_kernel, grid, block, 0, (hipStream_t)stream, c, h, w, n, fmap,
                           (long long)sb, (long long)sc, (long long)sh, xy, out);
    else
        hipLaunchKernelGGL(feature_gather_kernel, grid, block, 0, (hipStream_t)stream, c, h, w, n, fmap,
                           (long long)sb, (long long)sc, (long long)sh, (long long)sw, xy, out);
    return check_launch("feature_gather");
}

extern "C" int jm_feature_gather_grad(int b, int c, int h, int w, int n, const float* grad_out, const float* xy,
                                      float* grad_fmap, int64_t sb, int64_t sc, int64_t sh, int64_t sw,
                                      jm_stream_t stream) {
    JM_REQUIRE(b >= 0 && c >= 0 && h >= 1 && w >= 1 && n >= 0, "feature_gather_grad: bad sizes");
    if (b == 0 || c == 0 || n == 0) return JM_OK;
    JM_REQUIRE(grad_out && xy && grad_fmap, "feature_gather_grad: null pointer");
    JM_REQUIRE(b <= 65535 && divup(c, FG_CPT) <= 65535, "feature_gather_grad: shape too large");
    hipLaunchKernelGGL(feature_gather_grad_kernel, dim3(divup(n, 256), divup(c, FG_CPT), b), dim3(256), 0,
                       (hipStream_t)stream, c, h, w, n, grad_out, xy, grad_fmap, (long long)sb, (long long)sc,
                       (long long)sh, (long long)sw);
    return check_launch("feature_gather_grad");
}

/* training path: channels-last map (B, H, W, C) -> rows (B N, C) (ldo floats apart); same taps / products / summation order */
extern "C" int jm_feature_gather_rows(int b, int c, int h, int w, int n, const float* fmap_cl, const float* xy, float* out, int ldo,
                                      jm_stream_t stream) {
    JM_REQUIRE(b > 0 && c > 0 && c % 4 == 0 && h >= 1 && w >= 1 && n > 0 && ldo % 4 == 0 && ldo >= c && fmap_cl && xy && out,
               "feature_gather_rows: bad arguments (c %d and ldo %d must be multiples of 4)", c, ldo);
    JM_REQUIRE((reinterpret_cast<uintptr_t>(fmap_cl) & 15u) == 0 && (reinterpret_cast<uintptr_t>(xy) & 7u) == 0, "feature_gather_rows: alignment");
    const long long work = (long long)b * n * (c / 4);
    hipLaunchKernelGGL(feature_gather_rows_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, (hipStream_t)stream, b, c, h, w, n, fmap_cl, xy,
                       out, ldo);
    return check_launch("feature_gather_rows");
}

/* grad_fmap_cl (B, H, W, C) pre-zeroed by the caller (or holding a gradient to add to) */
extern "C" int jm_feature_gather_rows_grad(int b, int c, int h, int w, int n, const float* grad_out, int ldo, const float* xy, float* grad_fmap_cl,
                                           jm_stream_t stream) {
    JM_REQUIRE(b > 0 && c > 0 && h >= 1 && w >= 1 && n > 0 && ldo >= c && grad_out && xy && grad_fmap_cl, "feature_gather_rows_grad: bad arguments");
    const long long work = (long long)b * n * c;
    hipLaunchKernelGGL(feature_gather_rows_grad_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, (hipStream_t)stream, b, c, h, w, n,
                       grad_out, ldo, xy, grad_fmap_cl);
    return check_launch("feature_gather_rows_grad");
}
