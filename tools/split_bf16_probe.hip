// EXPERIMENTAL (never on the product path, never in a headline): can the fp32 GEMMs of the path (affinity
// heads, RCNN lift, 1x1 convolution chains; ~75 % of the MFMA time) run on the bf16 matrix pipe — 16x the fp32-MFMA rate —
// at fp32 accuracy?  Each fp32 operand is split into three bf16 terms a = a1 + a2 + a3 (8 + 8 + 8 significant bits, the
// residuals are exact in fp32), and the product a.b is evaluated as the SIX bf16 products of order <= 3
//     a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1)
// each exact in the MFMA's fp32 accumulator (8 x 8 bits); the three dropped terms are <= 2^-24 |a b| each, i.e. at the
// level of ONE fp32 rounding of the product.  6 x v_mfma_f32_32x32x16_bf16 (32 cycles each, 16 k) = 192 cycles per 16 k
// against 8 x v_mfma_f32_32x32x2_f32 (64 cycles each) = 512 cycles: 2.67 x fewer matrix-pipe cycles, before the VALU cost
// of the splitting (amortised over the tile when the split planes are staged once in LDS; here, without LDS, it is paid per
// use and the probe reports it as measured).
//
// Reports, for C = A B^T with M x K and N x K fp32 operands: max |error| vs an fp64 reference of (i) the exact-fp32 MFMA
// kernel, (ii) the 6-product split, (iii) a 3-product split (a1 b1 + a1 b2 + a2 b1: 16 bits), and the time of each.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/split_bf16_probe.hip -o tools/bin/split_bf16_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned short bf16_rne(float x) {
    unsigned u = __float_as_uint(x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

struct Split8 { bf16x8 p[3]; };

__device__ __forceinline__ Split8 split8(const float (&v)[8]) {
    Split8 s;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const unsigned short h1 = bf16_rne(v[i]);
        const float r1 = v[i] - bf16_f(h1);
        const unsigned short h2 = bf16_rne(r1);
        const float r2 = r1 - bf16_f(h2);
        const unsigned short h3 = bf16_rne(r2);
        s.p[0][i] = __builtin_bit_cast(__bf16, h1);
        s.p[1][i] = __builtin_bit_cast(__bf16, h2);
        s.p[2][i] = __builtin_bit_cast(__bf16, h3);
    }
    return s;
}

// one wave per 32 x 32 output tile, operands straight from L2.  MODE 0: exact fp32 MFMA; 1: six bf16 products; 2: three
template <int MODE>
__global__ void __launch_bounds__(64) gemm_probe(int M, int N, int K, const float* __restrict__ A, const float* __restrict__ B,
                                                 float* __restrict__ C) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
    const float* ap = A + (size_t)(m0 + r) * K;
    const float* bp = B + (size_t)(n0 + r) * K;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 16) {
        float a[8], b[8];
        const float4 a0 = *reinterpret_cast<const float4*>(ap + k0 + 8 * h), a1 = *reinterpret_cast<const float4*>(ap + k0 + 8 * h + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(bp + k0 + 8 * h), b1 = *reinterpret_cast<const float4*>(bp + k0 + 8 * h + 4);
        a[0] = a0.x; a[1] = a0.y; a[2] = a0.z; a[3] = a0.w; a[4] = a1.x; a[5] = a1.y; a[6] = a1.z; a[7] = a1.w;
        b[0] = b0.x; b[1] = b0.y; b[2] = b0.z; b[3] = b0.w; b[4] = b1.x; b[5] = b1.y; b[6] = b1.z; b[7] = b1.w;
        if (MODE == 0) {
            // lanes 0-31 hold k0 .. k0+7 of their row, lanes 32-63 k0+8 .. k0+15: step q pairs k0 + q with k0 + 8 + q
#pragma unroll
            for (int q = 0; q < 8; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], b[q], acc, 0, 0, 0);
        } else {
            const Split8 sa = split8(a), sb = split8(b);
            // smallest terms first
            if (MODE == 1) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa.p[0], sb.p[2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa.p[1], sb.p[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa.p[2], sb.p[0], acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa.p[0], sb.p[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa.p[1], sb.p[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa.p[0], sb.p[0], acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) C[(size_t)(m0 + (i & 3) + 8 * (i >> 2) + 4 * h) * N + n0 + r] = acc[i];
}

// the bare matrix-pipe rates: a dependent chain per wave, 4 waves per SIMD worth of work in flight
template <int MODE>
__global__ void __launch_bounds__(256) pipe_probe(int iters, float* out) {
    f32x16 acc[2];
    for (int j = 0; j < 2; ++j)
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    bf16x8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(0.5f + threadIdx.x * 1e-3f); y[i] = (__bf16)(1.0f); }
    const float fa = 0.5f + threadIdx.x * 1e-3f, fb = 1.f;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb, fa, acc[1], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, x, acc[1], 0, 0, 0);
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][5];
}

int main() {
    const int M = 16384, N = 512, K = 512;
    std::vector<float> A((size_t)M * K), B((size_t)N * K);
    srand(1);
    auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    for (auto& v : A) { v = rnd() * 2.f; if (v < 0) v = 0; }            // post-ReLU-like activations
    for (auto& v : B) v = rnd() * 0.1f;                                  // weights
    float *dA, *dB, *dC;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, (size_t)M * N * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    // fp64 reference on a sample of rows
    const int rows = 256;
    std::vector<double> ref((size_t)rows * N);
    double cmax = 0;
    for (int i = 0; i < rows; ++i)
        for (int j = 0; j < N; ++j) {
            double s = 0;
            for (int k = 0; k < K; ++k) s += (double)A[(size_t)(i * 61 % M) * K + k] * (double)B[(size_t)j * K + k];
            ref[(size_t)i * N + j] = s;
            cmax = fmax(cmax, fabs(s));
        }
    std::vector<float> C((size_t)M * N);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[3] = {"exact fp32 MFMA (v_mfma_f32_32x32x2_f32)", "split bf16, 6 products (orders 1-3)", "split bf16, 3 products (orders 1-2)"};
    for (int mode = 0; mode < 3; ++mode) {
        auto launch = [&] {
            dim3 grid(N / 32, M / 32);
            if (mode == 0) hipLaunchKernelGGL(gemm_probe<0>, grid, dim3(64), 0, 0, M, N, K, dA, dB, dC);
            else if (mode == 1) hipLaunchKernelGGL(gemm_probe<1>, grid, dim3(64), 0, 0, M, N, K, dA, dB, dC);
            else hipLaunchKernelGGL(gemm_probe<2>, grid, dim3(64), 0, 0, M, N, K, dA, dB, dC);
        };
        launch();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int it = 0; it < 10; ++it) launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        ms /= 10;
        hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
        double err = 0;
        for (int i = 0; i < rows; ++i)
            for (int j = 0; j < N; ++j) err = fmax(err, fabs((double)C[(size_t)(i * 61 % M) * N + j] - ref[(size_t)i * N + j]));
        printf("%-44s max|err| %.3e (|C|max %.2f, rel %.2e)   %.3f ms  %.1f TF (straight-from-L2 tiles: not a tuned GEMM)\n", names[mode], err,
               cmax, err / cmax, ms, 2.0 * M * N * K / ms / 1e9);
    }
    // bare pipe
    float* dout;
    hipMalloc(&dout, 1024 * 256 * 4);
    for (int mode = 0; mode < 2; ++mode) {
        const int iters = 2000;
        if (mode == 0) hipLaunchKernelGGL(pipe_probe<0>, dim3(1024), dim3(256), 0, 0, 10, dout); else hipLaunchKernelGGL(pipe_probe<1>, dim3(1024), dim3(256), 0, 0, 10, dout);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(pipe_probe<0>, dim3(1024), dim3(256), 0, 0, iters, dout); else hipLaunchKernelGGL(pipe_probe<1>, dim3(1024), dim3(256), 0, 0, iters, dout);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        // k = 16 of a 32 x 32 tile per `iteration` and accumulator: 2 accumulators x 4 waves x 1024 workgroups
        const double tiles = 2.0 * 4 * 1024 * iters;
        printf("bare pipe, %s: %.3f ms for %.0f tile-k16 steps = %.1f ns per 1000 steps; fp32-equivalent rate %.0f TF\n",
               mode == 0 ? "8 x mfma_f32_32x32x2_f32 per k16" : "6 x mfma_f32_32x32x16_bf16 per k16", ms, tiles, ms * 1e6 / tiles * 1000,
               tiles * 32 * 32 * 16 * 2 / ms / 1e9);
    }
    return 0;
}
