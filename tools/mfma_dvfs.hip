// tools/mfma_dvfs.hip — what the fp32 matrix pipe sustains on THIS box, and what an s_memtime tick is.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_dvfs.hip -o tools/bin/mfma_dvfs && tools/bin/mfma_dvfs
// Every wave issues v_mfma_f32_32x32x2_f32 back to back on four independent accumulators (no memory traffic inside
// the loop), one or two waves per SIMD, once on zero operands and once on random operands: the chip clocks to its
// power budget, so the same instruction stream runs at different frequencies (MI355X_MICROARCH.md, "DVFS give-back").
// Prints TFLOP/s from the wall clock, s_memtime ticks per MFMA, and the tick rate (ticks / wall time).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void __launch_bounds__(512) mfma_loop(const float* __restrict__ in, float* __restrict__ out,
                                                 unsigned long long* __restrict__ ticks, int iters) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    float a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = in[(gid * 16 + i) & 65535]; b[i] = in[(gid * 16 + 8 + i) & 65535]; }
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], b[kk], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], b[7 - kk], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[7 - kk], b[kk], acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[7 - kk], b[7 - kk], acc[3], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[gid] = s;
    if ((threadIdx.x & 63) == 0) ticks[gid >> 6] = t1 - t0;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("%s: %d CUs, clockRate %d kHz\n", prop.gcnArchName, cus, prop.clockRate);
    const int iters = 20000;
    std::vector<float> h(65536);
    float *in, *out; unsigned long long* ticks;
    CK(hipMalloc(&in, 65536 * 4)); CK(hipMalloc(&out, (size_t)cus * 512 * 4 * 4)); CK(hipMalloc(&ticks, (size_t)cus * 8 * 4 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int waves = 4; waves <= 8; waves += 4)
        for (int rnd = 0; rnd < 2; ++rnd) {
            srand(1);
            for (auto& v : h) v = rnd ? (float)rand() / RAND_MAX - 0.5f : 0.f;
            CK(hipMemcpy(in, h.data(), 65536 * 4, hipMemcpyHostToDevice));
            const int grid = cus * 2;          // two launches' worth of workgroups per CU in one grid keeps every CU busy
            for (int rep = 0; rep < 2; ++rep) {        // rep 0 warms the clocks
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(mfma_loop, dim3(grid), dim3(waves * 64), 0, 0, in, out, ticks, iters);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            }
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<unsigned long long> t((size_t)grid * waves);
            CK(hipMemcpy(t.data(), ticks, t.size() * 8, hipMemcpyDeviceToHost));
            double mean = 0; for (auto v : t) mean += (double)v; mean /= t.size();
            const double mfmas = (double)grid * waves * iters * 32.0;
            const double tf = mfmas * 32 * 32 * 2 * 2 / (ms * 1e-3) / 1e12;
            // waves per SIMD sharing the pipe: a CU holds 2 workgroups of 4 waves or 1 of 8 -> per-wave MFMA period
            printf("waves/WG %d  %s operands: %.3f ms  %.1f TFLOP/s  ticks per MFMA per wave %.1f  (wave lifetime %.3f ms -> %.0f MHz tick rate)\n",
                   waves, rnd ? "random" : "zero  ", ms, tf, mean / (iters * 32.0), ms / (grid * waves / (double)(cus * 8) > 1 ? grid * waves / (double)(cus * 8) : 1),
                   mean / (ms / ((grid * waves / (double)(cus * 8)) > 1 ? (grid * waves / (double)(cus * 8)) : 1) * 1e-3) / 1e6);
        }
    return 0;
}
