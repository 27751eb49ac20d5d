// fp32 MFMA issue-rate / clock probe: what does a loop of nothing but v_mfma_f32_32x32x2_f32 reach,
// and at what shader clock?   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/bin/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void __launch_bounds__(256) probe(float* out, long long* clk, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

template <int NACC>
void run(int wgs, int threads, int iters) {
    float* out; long long* clk;
    hipMalloc(&out, sizeof(float) * wgs * threads);
    hipMalloc(&clk, sizeof(long long) * 2 * wgs);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<NACC>, dim3(wgs), dim3(threads), 0, 0, out, clk, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<NACC>, dim3(wgs), dim3(threads), 0, 0, out, clk, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(2 * wgs);
    hipMemcpy(h.data(), clk, sizeof(long long) * 2 * wgs, hipMemcpyDeviceToHost);
    const double mf = (double)wgs * (threads / 64) * iters * 8.0 * NACC;
    const double tf = mf * 4096.0 / (ms * 1e-3) / 1e12;
    // wall_clock64 ticks at 100 MHz
    const double mhz = (double)h[0] / ((double)h[1] / 100.0);
    printf("wgs %4d x %3d thr, %d acc: %8.3f ms  %7.1f TF  shader clock %.0f MHz  cycles/MFMA/wave %.1f\n", wgs, threads,
           NACC, ms, tf, mhz, (double)h[0] / (iters * 8.0 * NACC));
    hipFree(out); hipFree(clk);
}

int main() {
    run<4>(256, 256, 20000);
    run<2>(256, 256, 40000);
    run<4>(256, 512, 10000);
    run<4>(2048, 256, 2500);
    run<4>(64, 256, 20000);
    return 0;
}
