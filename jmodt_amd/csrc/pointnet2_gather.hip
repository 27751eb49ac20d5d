// pointnet2_gather.hip — gather / group / three_nn / three_interpolate (+ grads) for gfx950.
//
// Replaces gather_points_* (sampling_gpu.cu:8-83), group_points_* (group_points_gpu.cu:8-86),
// three_nn / three_interpolate(_grad) (interpolate_gpu.cu:9-161) of jmodt/ops/pointnet2/src.
//
// Common shape of the gather kernels: the index is read ONCE per output position and reused
// for a block of channels (the reference re-reads idx for every channel: grid.y = C), output
// stores are contiguous along the fastest axis and 16 bytes wide where alignment allows.
#include "jm_common.h"

namespace jm {

// ------------------------------------------------------------------ gather_points
__global__ void gather_points_kernel(int c, int n, int m, const float* __restrict__ points,
                                     const int* __restrict__ idx, float* __restrict__ out) {
    const int bi = blockIdx.z;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const int src = idx[(size_t)bi * m + j];
    const int c0 = blockIdx.y * 8;
    const int c1 = min(c, c0 + 8);
    for (int ci = c0; ci < c1; ++ci)
        out[((size_t)bi * c + ci) * m + j] = points[((size_t)bi * c + ci) * n + src];
}

__global__ void gather_points_grad_kernel(int c, int n, int m, const float* __restrict__ grad_out,
                                          const int* __restrict__ idx, float* __restrict__ grad_points) {
    const int bi = blockIdx.z;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const int dst = idx[(size_t)bi * m + j];
    const int c0 = blockIdx.y * 8;
    const int c1 = min(c, c0 + 8);
    for (int ci = c0; ci < c1; ++ci)
        unsafeAtomicAdd(grad_points + ((size_t)bi * c + ci) * n + dst, grad_out[((size_t)bi * c + ci) * m + j]);
}

// ------------------------------------------------------------------ group_points
// out[b,c,q] = points[b,c,idx[b,q]],  q = p*nsample + s  (contiguous).  Each thread owns 4
// consecutive q (one int4 index load, one float4 store per channel) and CPT channels.
constexpr int GP_CPT = 8;

template <bool VEC4>
__global__ void __launch_bounds__(256)
group_points_kernel(int c, int n, int q_total, const float* __restrict__ points, const int* __restrict__ idx,
                    float* __restrict__ out) {
    const int bi = blockIdx.z;
    const int c0 = blockIdx.y * GP_CPT;
    const int c1 = min(c, c0 + GP_CPT);
    const int* ix = idx + (size_t)bi * q_total;
    if (VEC4) {
        const int q = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
        if (q >= q_total) return;
        const int4 i4 = *reinterpret_cast<const int4*>(ix + q);
#pragma unroll 4
        for (int ci = c0; ci < c1; ++ci) {
            const float* src = points + ((size_t)bi * c + ci) * n;
            float4 v;
            v.x = src[i4.x]; v.y = src[i4.y]; v.z = src[i4.z]; v.w = src[i4.w];
            // streaming store: the grouped tensor (100 MB at level 2) is written once and read once, much later
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            __builtin_nontemporal_store((f32x4){v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4*>(out + ((size_t)bi * c + ci) * q_total + q));
        }
    } else {
        const int q = blockIdx.x * blockDim.x + threadIdx.x;
        if (q >= q_total) return;
        const int i = ix[q];
        for (int ci = c0; ci < c1; ++ci)
            out[((size_t)bi * c + ci) * q_total + q] = points[((size_t)bi * c + ci) * n + i];
    }
}

// Backward of the grouping: grad_points[b][c][idx[b][q]] += grad_out[b][c][q] (group_points_gpu.cu:48-86: one atomicAdd per
// element).  A ball-query list ends in copies of its first hit (ball_query_gpu.cu:36-40) — on the FPS-thinned RPN clouds 9 of 10
// slots — so most of a wave's 64 consecutive q would add to the SAME address, which the L2 atomic unit serialises.  Each RUN of
// equal consecutive indices inside a wave is therefore summed in registers first (segmented shuffle scan, run bounds from one ballot
// of the head flags) and only its last lane issues the atomic.  (Float atomics: the summation order was never fixed.)
__global__ void __launch_bounds__(256)
group_points_grad_kernel(int c, int n, int q_total, const float* __restrict__ grad_out,
                         const int* __restrict__ idx, float* __restrict__ grad_points) {
    const int bi = blockIdx.z;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool live = q < q_total;
    const int i = live ? idx[(size_t)bi * q_total + q] : -1 - lane;            // dead lanes: runs of their own, never stored
    const int prev = __shfl_up(i, 1);
    const unsigned long long heads = __ballot(lane == 0 || i != prev);
    const int start = 63 - __builtin_clzll(heads & (~0ull >> (63 - lane)));     // first lane of this lane's run
    const bool tail = lane == 63 || ((heads >> (lane + 1)) & 1ull);
    const int c0 = blockIdx.y * GP_CPT;
    const int c1 = min(c, c0 + GP_CPT);
    if (heads == ~0ull) {                                                       // no two neighbours alike: the plain form
        if (live)
            for (int ci = c0; ci < c1; ++ci)
                unsafeAtomicAdd(grad_points + ((size_t)bi * c + ci) * n + i, grad_out[((size_t)bi * c + ci) * q_total + q]);
        return;
    }
    for (int ci = c0; ci < c1; ++ci) {
        float v = live ? grad_out[((size_t)bi * c + ci) * q_total + q] : 0.f;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const float up = __shfl_up(v, d);
            if (lane - d >= start) v += up;
        }
        if (live && tail) unsafeAtomicAdd(grad_points + ((size_t)bi * c + ci) * n + i, v);
    }
}

// ------------------------------------------------------------------ three_nn
// Lane = unknown point; the known point is wave-uniform and arrives through scalar loads in
// groups of 8 with two named buffers (same structure as ball_query.hip).  The scan of the known
// set is split over the 4 waves of a workgroup (wave w scans the w-th contiguous slice); the four
// partial top-3 lists are merged in slice order with the same strict `<` insertion, which yields
// exactly the 3 smallest by (distance, index) — what the sequential scan of
// interpolate_gpu.cu:30-51 produces.  Within a group of 8 the insertion logic only runs when some
// lane of the wave has a candidate closer than its current third best.
// best* are kept in float with +inf start, which is decision-for-decision identical to the
// reference's double bests initialised to 1e40 (every finite float d satisfies d < 1e40 and
// d < inf alike; inf < either is false).
struct Top3 {
    float b1, b2, b3;
    int i1, i2, i3;
    __device__ __forceinline__ void visit(int k, float d) {
        const bool c1 = d < b1, c2 = d < b2, c3 = d < b3;
        // if/else-if chain of interpolate_gpu.cu:37-49, branch-free (c1 => c2 => c3)
        b3 = c2 ? b2 : (c3 ? d : b3);  i3 = c2 ? i2 : (c3 ? k : i3);
        b2 = c1 ? b1 : (c2 ? d : b2);  i2 = c1 ? i1 : (c2 ? k : i2);
        b1 = c1 ? d : b1;              i1 = c1 ? k : i1;
    }
};

__global__ void __launch_bounds__(256)
three_nn_kernel(int n, int m, int chunk, const float* __restrict__ unknown, const float* __restrict__ known,
                float* __restrict__ dist2, int* __restrict__ idx) {
    __shared__ float sd[4][3][64];
    __shared__ int si[4][3][64];
    const int bi = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pi = blockIdx.x * 64 + lane;
    const bool active = pi < n;
    const float* u = unknown + ((size_t)bi * n + (active ? pi : 0)) * 3;
    const float ux = u[0], uy = u[1], uz = u[2];
    const float* kn = known + (size_t)bi * m * 3;
    Top3 t{INFINITY, INFINITY, INFINITY, 0, 0, 0};
    const int k_begin = wave * chunk, k_end = min(m, k_begin + chunk);
    int k = k_begin;
    if ((reinterpret_cast<uintptr_t>(kn) & 15u) == 0 && k + 16 <= k_end) {
        auto load8 = [&](int kk, float4 (&q)[6]) {
            const float4* src = reinterpret_cast<const float4*>(kn + (size_t)kk * 3);  // wave-uniform
#pragma unroll
            for (int i = 0; i < 6; ++i) q[i] = src[i];
        };
        auto eval8 = [&](int kk, const float4 (&q)[6]) {
            const float f[24] = {q[0].x, q[0].y, q[0].z, q[0].w, q[1].x, q[1].y, q[1].z, q[1].w,
                                 q[2].x, q[2].y, q[2].z, q[2].w, q[3].x, q[3].y, q[3].z, q[3].w,
                                 q[4].x, q[4].y, q[4].z, q[4].w, q[5].x, q[5].y, q[5].z, q[5].w};
            float d[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) d[i] = sqdist3(ux - f[3 * i], uy - f[3 * i + 1], uz - f[3 * i + 2]);
            const float dmin = fminf(fminf(fminf(d[0], d[1]), fminf(d[2], d[3])), fminf(fminf(d[4], d[5]), fminf(d[6], d[7])));
            if (__any(dmin < t.b3)) {
                // only the points that improve SOME lane's list run the insertion (uniform branch per point;
                // a point no lane wants leaves every list unchanged, so skipping it is exact)
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (__any(d[i] < t.b3)) t.visit(kk + i, d[i]);
            }
        };
        float4 ga[6], gb[6];
        load8(k, ga);
        load8(k + 8, gb);
        for (; k + 32 <= k_end; k += 16) {
            eval8(k, ga);
            load8(k + 16, ga);
            eval8(k + 8, gb);
            load8(k + 24, gb);
        }
        eval8(k, ga);
        eval8(k + 8, gb);
        k += 16;
    }
    for (; k < k_end; ++k) t.visit(k, sqdist3(ux - kn[k * 3 + 0], uy - kn[k * 3 + 1], uz - kn[k * 3 + 2]));
    sd[wave][0][lane] = t.b1; sd[wave][1][lane] = t.b2; sd[wave][2][lane] = t.b3;
    si[wave][0][lane] = t.i1; si[wave][1][lane] = t.i2; si[wave][2][lane] = t.i3;
    __syncthreads();
    if (wave == 0 && active) {
        Top3 r{INFINITY, INFINITY, INFINITY, 0, 0, 0};
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
            for (int q = 0; q < 3; ++q) r.visit(si[w][q][lane], sd[w][q][lane]);
        float* d = dist2 + ((size_t)bi * n + pi) * 3;
        int* o = idx + ((size_t)bi * n + pi) * 3;
        d[0] = r.b1; d[1] = r.b2; d[2] = r.b3;
        o[0] = r.i1; o[1] = r.i2; o[2] = r.i3;
    }
}

// ------------------------------------------------------------------ three_interpolate
constexpr int TI_CPT = 8;

__global__ void __launch_bounds__(256)
three_interpolate_kernel(int c, int m, int n, const float* __restrict__ points, const int* __restrict__ idx,
                         const float* __restrict__ weight, float* __restrict__ out) {
    const int bi = blockIdx.z;
    const int pi = blockIdx.x * blockDim.x + threadIdx.x;
    if (pi >= n) return;
    const int* ix = idx + ((size_t)bi * n + pi) * 3;
    const float* w = weight + ((size_t)bi * n + pi) * 3;
    const int i0 = ix[0], i1 = ix[1], i2 = ix[2];
    const float w0 = w[0], w1 = w[1], w2 = w[2];
    const int c0 = blockIdx.y * TI_CPT;
    const int c1 = min(c, c0 + TI_CPT);
#pragma unroll 4
    for (int ci = c0; ci < c1; ++ci) {
        const float* src = points + ((size_t)bi * c + ci) * m;
        // w0*p0 + w1*p1 + w2*p2 with the oracle's contraction
        out[((size_t)bi * c + ci) * n + pi] = __builtin_fmaf(w2, src[i2], __builtin_fmaf(w0, src[i0], w1 * src[i1]));
    }
}

// LDS-staged variant for the forward pass.  The plain kernel is bound by the texture addresser:
// every wave-level gather touches ~64 different cache lines of the (b,c) feature row, 3 per output.
// Here a workgroup stages CB whole feature rows (CB * m floats) in LDS once, then every thread
// walks points: idx / weights are read ONCE per point for all CB channels and the 3 taps per
// channel are LDS reads.  One workgroup per (batch, channel block): with C = 256, m = 4096 that is
// 8 rows = 128 KiB of LDS and exactly 256 workgroups.
__global__ void __launch_bounds__(512)
three_interpolate_lds_kernel(int c, int m, int n, int cb, const float* __restrict__ points,
                             const int* __restrict__ idx, const float* __restrict__ weight, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float rows[];   // [cb][m]
    const int bi = blockIdx.y;
    const int c0 = blockIdx.x * cb;
    const int nc = min(cb, c - c0);
    const float* src = points + ((size_t)bi * c + c0) * m;          // nc consecutive rows are contiguous
    const int tot = nc * m;
    if (((reinterpret_cast<uintptr_t>(src) & 15u) == 0) && (tot % 4 == 0)) {
        for (int e = threadIdx.x; e < tot / 4; e += blockDim.x)
            reinterpret_cast<float4*>(rows)[e] = reinterpret_cast<const float4*>(src)[e];
    } else {
        for (int e = threadIdx.x; e < tot; e += blockDim.x) rows[e] = src[e];
    }
    __syncthreads();
    const int* ixb = idx + (size_t)bi * n * 3;
    const float* wb = weight + (size_t)bi * n * 3;
    float* ob = out + ((size_t)bi * c + c0) * n;
    for (int pi = threadIdx.x; pi < n; pi += blockDim.x) {
        const int i0 = ixb[pi * 3 + 0], i1 = ixb[pi * 3 + 1], i2 = ixb[pi * 3 + 2];
        const float w0 = wb[pi * 3 + 0], w1 = wb[pi * 3 + 1], w2 = wb[pi * 3 + 2];
        for (int cc = 0; cc < nc; ++cc) {
            const float* r = rows + cc * m;
            ob[(size_t)cc * n + pi] = __builtin_fmaf(w2, r[i2], __builtin_fmaf(w0, r[i0], w1 * r[i1]));
        }
    }
}

__global__ void __launch_bounds__(256)
three_interpolate_grad_kernel(int c, int n, int m, const float* __restrict__ grad_out,
                              const int* __restrict__ idx, const float* __restrict__ weight,
                              float* __restrict__ grad_points) {
    const int bi = blockIdx.z;
    const int pi = blockIdx.x * blockDim.x + threadIdx.x;
    if (pi >= n) return;
    const int* ix = idx + ((size_t)bi * n + pi) * 3;
    const float* w = weight + ((size_t)bi * n + pi) * 3;
    const int i0 = ix[0], i1 = ix[1], i2 = ix[2];
    const float w0 = w[0], w1 = w[1], w2 = w[2];
    const int c0 = blockIdx.y * TI_CPT;
    const int c1 = min(c, c0 + TI_CPT);
    for (int ci = c0; ci < c1; ++ci) {
        const float g = grad_out[((size_t)bi * c + ci) * n + pi];
        float* dst = grad_points + ((size_t)bi * c + ci) * m;
        unsafeAtomicAdd(dst + i0, g * w0);
        unsafeAtomicAdd(dst + i1, g * w1);
        unsafeAtomicAdd(dst + i2, g * w2);
    }
}

}  // namespace jm

using namespace jm;

extern "C" int jm_gather_points(int b, int c, int n, int npoints, const float* points, const int* idx, float* out,
                                jm_stream_t stream) {
    JM_REQUIRE(b >= 0 && c >= 0 && n >= 0 && npoints >= 0, "gather_points: bad sizes");
    if (b == 0 || c == 0 || npoints == 0) return JM_OK;
    JM_REQUIRE(points && idx && out, "gather_points: null pointer");
    JM_REQUIRE(b <= 65535, "gather_points: batch %d > 65535", b);
    hipLaunchKernelGGL(gather_points_kernel, dim3(divup(npoints, 256), divup(c, 8), b), dim3(256), 0,
                       (hipStream_t)stream, c, n, npoints, points, idx, out);
    return check_launch("gather_points");
}

extern "C" int jm_gather_points_grad(int b, int c, int n, int npoints, const float* grad_out, const int* idx,
                                     float* grad_points, jm_stream_t stream) {
    JM_REQUIRE(b >= 0 && c >= 0 && n >= 0 && npoints >= 0, "gather_points_grad: bad sizes");
    if (b == 0 || c == 0 || npoints == 0) return JM_OK;
    JM_REQUIRE(grad_out && idx && grad_points, "gather_points_grad: null pointer");
    JM_REQUIRE(b <= 65535, "gather_points_grad: batch %d > 65535", b);
    hipLaunchKernelGGL(gather_points_grad_kernel, dim3(divup(npoints, 256), divup(c, 8), b), dim3(256), 0,
                       (hipStream_t)stream, c, n, npoints, grad_out, idx, grad_points);
    return check_launch("gather_points_grad");
}

extern "C" int jm_group_points(int b, int c, int n, int npoints, int nsample, const float* points, const int* idx,
                               float* out, jm_stream_t stream) {
    JM_REQUIRE(b >= 0 && c >= 0 && n >= 0 && npoints >= 0 && nsample >= 0, "group_points: bad sizes");
    const long long qt = (long long)npoints * nsample;
    if (b == 0 || c == 0 || qt == 0) return JM_OK;
    JM_REQUIRE(points && idx && out, "group_points: null pointer");
    JM_REQUIRE(b <= 65535 && divup(c, GP_CPT) <= 65535 && qt < (1LL << 31), "group_points: shape too large");
    const int q_total = (int)qt;
    const bool vec = (q_total % 4 == 0) && ((reinterpret_cast<uintptr_t>(idx) & 15u) == 0) &&
                     ((reinterpret_cast<uintptr_t>(out) & 15u) == 0);
    if (vec)
        hipLaunchKernelGGL(group_points_kernel<true>, dim3(divup(q_total / 4, 256), divup(c, GP_CPT), b), dim3(256), 0,
                           (hipStream_t)stream, c, n, q_total, points, idx, out);
    else
        hipLaunchKernelGGL(group_points_kernel<false>, dim3(divup(q_total, 256), divup(c, GP_CPT), b), dim3(256), 0,
                           (hipStream_t)stream, c, n, q_total, points, idx, out);
    return check_launch("group_points");
}

extern "C" int jm_group_points_grad(int b, int c, int n, int npoints, int nsample, const float* grad_out,
                                    const int* idx, float* grad_points, jm_stream_t stream) {
    JM_REQUIRE(b >= 0 && c >= 0 && n >= 0 && npoints >= 0 && nsample >= 0, "group_points_grad: bad sizes");
    const long long qt = (long long)npoints * nsample;
    if (b == 0 || c == 0 || qt == 0) return JM_OK;
    JM_REQUIRE(grad_out && idx && grad_points, "group_points_grad: null pointer");
    JM_REQUIRE(b <= 65535 && divup(c, GP_CPT) <= 65535 && qt < (1LL << 31), "group_points_grad: shape too large");
    hipLaunchKernelGGL(group_points_grad_kernel, dim3(divup((int)qt, 256), divup(c, GP_CPT), b), dim3(256), 0,
                       (hipStream_t)stream, c, n, (int)qt, grad_out, idx, grad_points);
    return check_launch("group_points_grad");
}

extern "C" int jm_three_nn(int b, int n, int m, const float* unknown, const float* known, float* dist2, int* idx,
                           jm_stream_t stream) {
    JM_REQUIRE(b >= 0 && n >= 0 && m >= 0, "three_nn: bad sizes");
    if (b == 0 || n == 0) return JM_OK;
    JM_REQUIRE(unknown && dist2 && idx && (known || m == 0), "three_nn: null pointer");
    JM_REQUIRE(b <= 65535, "three_nn: batch %d > 65535", b);
    const int chunk = (divup(m, 4) + 7) / 8 * 8;   // slice per wave, multiple of 8 (16-byte aligned groups)
    hipLaunchKernelGGL(three_nn_kernel, dim3(divup(n, 64), b), dim3(256), 0, (hipStream_t)stream, n, m, chunk,
                       unknown, known, dist2, idx);
    return check_launch("three_nn");
}

extern "C" int jm_three_interpolate(int b, int c, int m, int n, const float* points, const int* idx,
                                    const float* weight, float* out, jm_stream_t stream) {
    JM_REQUIRE(b >= 0 && c >= 0 && m >= 0 && n >= 0, "three_interpolate: bad sizes");
    if (b == 0 || c == 0 || n == 0) return JM_OK;
    JM_REQUIRE(points && idx && weight && out, "three_interpolate: null pointer");
    JM_REQUIRE(b <= 65535 && divup(c, TI_CPT) <= 65535, "three_interpolate: shape too large");
    // LDS-staged path when a useful block of feature rows fits in LDS and there are enough
    // points per row to amortise the staging
    const size_t row_bytes = (size_t)m * sizeof(float);
    int cb = (int)((128u * 1024u) / (row_bytes ? row_bytes : 1));
    if (cb > 8) cb = 8;
    if (cb > c) cb = c;
    if (cb >= 1 && n >= 4 * m && m >= 64) {
        while (cb > 1 && (long long)b * divup(c, cb) < 256) cb /= 2;   // keep >= 256 workgroups when possible
        const size_t lds = (size_t)cb * row_bytes;
        if (lds > 64 * 1024)
            (void)hipFuncSetAttribute((const void*)three_interpolate_lds_kernel,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(three_interpolate_lds_kernel, dim3(divup(c, cb), b), dim3(512), lds, (hipStream_t)stream,
                           c, m, n, cb, points, idx, weight, out);
        return check_launch("three_interpolate(lds)");
    }
    hipLaunchKernelGGL(three_interpolate_kernel, dim3(divup(n, 256), divup(c, TI_CPT), b), dim3(256), 0,
                       (hipStream_t)stream, c, m, n, points, idx, weight, out);
    return check_launch("three_interpolate");
}

extern "C" int jm_three_interpolate_grad(int b, int c, int n, int m, const float* grad_out, const int* idx,
                                         const float* weight, float* grad_points, jm_stream_t stream) {
    JM_REQUIRE(b >= 0 && c >= 0 && m >= 0 && n >= 0, "three_interpolate_grad: bad sizes");
    if (b == 0 || c == 0 || n == 0) return JM_OK;
    JM_REQUIRE(grad_out && idx && weight && grad_points, "three_interpolate_grad: null pointer");
    JM_REQUIRE(b <= 65535 && divup(c, TI_CPT) <= 65535, "three_interpolate_grad: shape too large");
    hipLaunchKernelGGL(three_interpolate_grad_kernel, dim3(divup(n, 256), divup(c, TI_CPT), b), dim3(256), 0,
                       (hipStream_t)stream, c, n, m, grad_out, idx, weight, grad_points);
    return check_launch("three_interpolate_grad");
}
