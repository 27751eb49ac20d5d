// what slows an MFMA stream down?  variants of the fused-SA inner block on one workgroup per CU
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int THREADS = 256>
__global__ void __launch_bounds__(THREADS) probe(const float* __restrict__ W, float* out, long long* clk, int iters) {
    __shared__ float A[128 * 132];
    for (int i = threadIdx.x; i < 128 * 132; i += THREADS) A[i] = i * 1e-4f;
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3, lr = lane & 31, lk = lane >> 5;
    const int a_off = lk * 132 + (wave >> 1) * 64 + lr;
    const float* bp = W + ((size_t)(wave & 1) * 64 + lr) * 16 + lk * 8;
    float4 bc[4], bn[4];
    float ac[16], an[16];
    auto loadB = [&](float4 (&b)[4], const float* q) {
        b[0] = *reinterpret_cast<const float4*>(q); b[1] = *reinterpret_cast<const float4*>(q + 4);
        b[2] = *reinterpret_cast<const float4*>(q + 512); b[3] = *reinterpret_cast<const float4*>(q + 516);
    };
    auto loadA = [&](float (&a)[16], int kt) {
        const float* q = A + kt * 16 * 132 + a_off;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) { a[2 * kk] = q[(2 * kk) * 132]; a[2 * kk + 1] = q[(2 * kk) * 132 + 32]; }
    };
    float vsub[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) vsub[q] = A[q * 132 + lane] * 1e-3f;
    auto mm = [&](const float (&a0)[16], const float4 (&b)[4]) {
        float a[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) a[q] = (MODE & 16) ? fmaxf(a0[q] - vsub[q], 0.f) : a0[q];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const float b0 = reinterpret_cast<const float*>(&b[0])[kk], b1 = reinterpret_cast<const float*>(&b[2])[kk];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * kk], b0, acc[0][0], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * kk + 1], b0, acc[1][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * kk], b1, acc[0][1], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * kk + 1], b1, acc[1][1], 0, 0, 0);
        }
    };
    loadB(bc, bp); loadA(ac, 0); loadB(bn, bp + 2048); loadA(an, 1);
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        const int kt = (it * 2) & 7;
        if (MODE & 1) loadB(bn, bp + (size_t)(kt + 1) * 2048);
        if (MODE & 2) loadA(an, kt + 1);
        __builtin_amdgcn_sched_barrier(0);
        mm(ac, bc);
        __builtin_amdgcn_sched_barrier(0);
        if (MODE & 1) loadB(bc, bp + (size_t)((kt + 2) & 7) * 2048);
        if (MODE & 2) loadA(ac, (kt + 2) & 7);
        __builtin_amdgcn_sched_barrier(0);
        mm(an, bn);
        __builtin_amdgcn_sched_barrier(0);
        if (MODE & 4) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if ((MODE & 8) && (it & 3) == 3) {     // a stage boundary every 8 k-tiles: hidden epilogue + zero + barrier
            float* Y = A + ((it >> 2) & 1 ? 0 : 64 * 132);
            const int wm = wave >> 1, wn = wave & 1;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {
                        float4 v;
                        v.x = fmaxf(acc[i][j][4 * rq + 0] + 0.5f, 0.f); v.y = fmaxf(acc[i][j][4 * rq + 1] + 0.5f, 0.f);
                        v.z = fmaxf(acc[i][j][4 * rq + 2] + 0.5f, 0.f); v.w = fmaxf(acc[i][j][4 * rq + 3] + 0.5f, 0.f);
                        *reinterpret_cast<float4*>(Y + (size_t)((wn * 32 + j * 16 + (lr >> 1))) * 132 + wm * 64 + i * 32 + 8 * rq + 4 * lk) = v;
                    }
            for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            loadA(ac, 0);
        }
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

template <int MODE, int THREADS = 256>
void run(const char* what) {
    const int wgs = 256, iters = 4000;
    float *out, *W; long long* clk;
    hipMalloc(&out, sizeof(float) * wgs * THREADS); hipMalloc(&clk, sizeof(long long) * 2 * wgs);
    hipMalloc(&W, sizeof(float) * 2048 * 16); hipMemset(W, 0, sizeof(float) * 2048 * 16);
    hipLaunchKernelGGL((probe<MODE, THREADS>), dim3(wgs), dim3(THREADS), 0, 0, W, out, clk, 10);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<MODE, THREADS>), dim3(wgs), dim3(THREADS), 0, 0, W, out, clk, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(2 * wgs);
    hipMemcpy(h.data(), clk, sizeof(long long) * 2 * wgs, hipMemcpyDeviceToHost);
    printf("%-52s %d waves/SIMD: cycles per MFMA per SIMD %.1f   clock %.0f MHz  %.1f TF\n", what, THREADS / 256, (double)h[0] / (iters * 64.0) / (THREADS / 256), (double)h[0] / (h[1] / 100.0), (double)wgs * (THREADS / 64) * iters * 64.0 * 4096 / (ms * 1e-3) / 1e12);
}

int main() {
    run<0>("mm only (operands fixed in 48 regs)");
    run<2>("+ A from LDS each k-tile");
    run<1>("+ B from global each k-tile");
    run<3>("+ both");
    run<7>("+ both + barrier per 2 k-tiles");
    run<11>("+ both + stage boundary per 8 k-tiles");
    run<19>("+ both + relu(a - v) VALU on the A operand");
    run<27>("+ both + VALU + stage boundary");
    run<0, 512>("mm only");
    run<3, 512>("+ both feeds");
    run<11, 512>("+ both + stage boundary per 8 k-tiles");
    run<19, 512>("+ both + relu(a - v) VALU on the A operand");
    run<27, 512>("+ both + VALU + stage boundary");
    return 0;
}
