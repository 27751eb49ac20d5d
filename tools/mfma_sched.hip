// tools/mfma_sched.hip — scheduling variants of the fused SA kernel's hoisted-layer inner block (jm_mfma.h, mfma_ktiles
// <TWO, SUBV>): one MFMA wave per SIMD, A operand = relu(a - v) with a from a k-major LDS tile and v from an LDS table,
// B operand from global (L2) in the packed layout.   hipcc --offload-arch=gfx950 -O3 -w tools/mfma_sched.hip -o tools/bin/mfma_sched
//   V0  what the kernel does today: loads of the next k-tile in one clump, relu(a - v) right before each MFMA
//   V1  V0 with the loads spread between the MFMAs (sched_group_barrier)
//   V2  relu(a - v) of the NEXT k-tile computed under the MFMAs of the current one (no VALU -> MFMA dependency in flight)
//   V3  V2 with loads and VALU spread evenly (sched_group_barrier)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define SB() __builtin_amdgcn_sched_barrier(0)
#define SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)

template <int V>
__global__ void __launch_bounds__(256) probe(const float* __restrict__ W, float* out, int iters) {
    __shared__ float A[128 * 132];
    __shared__ float VT[2 * 8 * 16];
    for (int i = threadIdx.x; i < 128 * 132; i += 256) A[i] = (i % 977) * 1e-3f;
    for (int i = threadIdx.x; i < 256; i += 256) VT[i] = (i % 13) * 0.05f;
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 31, lk = lane >> 5;
    const int a_off = lk * 132 + (wave >> 1) * 64 + lr;
    const float* bp = W + ((size_t)(wave & 1) * 64 + lr) * 16 + lk * 8;
    const float* vt0 = VT + lk * 8, *vt1 = VT + 128 + lk * 8;
    auto loadB = [&](float4 (&b)[4], int kt) __attribute__((always_inline)) {
        const float* q = bp + (size_t)(kt & 7) * 2048;
        b[0] = *reinterpret_cast<const float4*>(q); b[1] = *reinterpret_cast<const float4*>(q + 4);
        b[2] = *reinterpret_cast<const float4*>(q + 512); b[3] = *reinterpret_cast<const float4*>(q + 516);
    };
    auto loadA = [&](float (&a)[16], int kt) __attribute__((always_inline)) {
        const float* q = A + (kt & 7) * 16 * 132 + a_off;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) { a[2 * kk] = q[(2 * kk) * 132]; a[2 * kk + 1] = q[(2 * kk) * 132 + 32]; }
    };
    auto loadV = [&](float4 (&v)[4], int kt) __attribute__((always_inline)) {
        v[0] = *reinterpret_cast<const float4*>(vt0 + (kt & 7) * 16); v[1] = *reinterpret_cast<const float4*>(vt0 + (kt & 7) * 16 + 4);
        v[2] = *reinterpret_cast<const float4*>(vt1 + (kt & 7) * 16); v[3] = *reinterpret_cast<const float4*>(vt1 + (kt & 7) * 16 + 4);
    };
    auto relu_sub = [&](float (&t)[16], const float (&a)[16], const float4 (&v)[4]) __attribute__((always_inline)) {
        const float v0[8] = {v[0].x, v[0].y, v[0].z, v[0].w, v[1].x, v[1].y, v[1].z, v[1].w};
        const float v1[8] = {v[2].x, v[2].y, v[2].z, v[2].w, v[3].x, v[3].y, v[3].z, v[3].w};
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) { t[2 * kk] = fmaxf(a[2 * kk] - v0[kk], 0.f); t[2 * kk + 1] = fmaxf(a[2 * kk + 1] - v1[kk], 0.f); }
    };
    auto copy16 = [&](float (&t)[16], const float (&a)[16]) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 16; ++q) t[q] = a[q];
    };
    auto mm_raw = [&](const float (&a)[16], const float4 (&b)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const float b0 = reinterpret_cast<const float*>(&b[0])[kk], b1 = reinterpret_cast<const float*>(&b[2])[kk];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * kk], b0, acc[0][0], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * kk + 1], b0, acc[1][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * kk], b1, acc[0][1], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * kk + 1], b1, acc[1][1], 0, 0, 0);
        }
    };
    auto spread = [&]() __attribute__((always_inline)) {       // 32 MFMAs, one other instruction or two after each
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            SGB(0x008, 1);
            if (i < 12) SGB(0x100, 1);
            else if (i < 16) SGB(0x020, 1);
            SGB(0x002, V >= 2 ? 2 : 1);
        }
    };
    if (V < 2) {
        float4 bc[4], bn[4], vc[4], vn[4];
        float ac[16], an[16], t[16];
        loadB(bc, 0); loadA(ac, 0); loadV(vc, 0);
        for (int it = 0; it < iters; ++it) {
            const int kt = it * 2;
            loadB(bn, kt + 1); loadA(an, kt + 1); loadV(vn, kt + 1);
            if (V == 0) SB();
            relu_sub(t, ac, vc); mm_raw(t, bc);
            if (V == 1) spread();
            SB();
            loadB(bc, kt + 2); loadA(ac, kt + 2); loadV(vc, kt + 2);
            if (V == 0) SB();
            relu_sub(t, an, vn); mm_raw(t, bn);
            if (V == 1) spread();
            SB();
        }
    } else {
        float4 b0[4], b1[4], v0[4], v1[4];
        float r0[16], r1[16], t0[16], t1[16];
        // stage s consumes t[s & 1] and b[s & 1]; converts r[(s+1) & 1], v[(s+1) & 1] -> t[(s+1) & 1]; loads r[s & 1], v[s & 1] (stage s+2), b[(s+1)&1] (stage s+1)
        loadA(r0, 0); loadV(v0, 0); loadB(b0, 0);
        loadA(r1, 1); loadV(v1, 1);
        relu_sub(t0, r0, v0);
        for (int it = 0; it < iters; ++it) {
            const int s = it * 2;
            loadB(b1, s + 1); loadA(r0, s + 2); loadV(v0, s + 2);     // r0 / v0 were consumed when t0 was formed
            if (V == 2 || V == 5) SB();
            if (V == 5) copy16(t1, r1); else relu_sub(t1, r1, v1); if (V == 4) SB(); mm_raw(t0, b0);
            if (V == 3) spread();
            SB();
            loadB(b0, s + 2); loadA(r1, s + 3); loadV(v1, s + 3);
            if (V == 2 || V == 5) SB();
            if (V == 5) copy16(t0, r0); else relu_sub(t0, r0, v0); if (V == 4) SB(); mm_raw(t1, b1);
            if (V == 3) spread();
            SB();
        }
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int V>
void run(const char* what) {
    const int wgs = 256, iters = 4000;
    float *out, *W;
    hipMalloc(&out, sizeof(float) * wgs * 256);
    hipMalloc(&W, sizeof(float) * 2048 * 16);
    std::vector<float> h(2048 * 16);
    for (size_t i = 0; i < h.size(); ++i) h[i] = ((i * 2654435761u) % 1000) * 1e-4f - 0.05f;
    hipMemcpy(W, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe<V>, dim3(wgs), dim3(256), 0, 0, W, out, 10);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<V>, dim3(wgs), dim3(256), 0, 0, W, out, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)iters * 64.0;
    printf("%-72s %.3f ms  %.1f cycles/MFMA @2.4GHz  %.1f TF\n", what, ms, ms * 1e-3 * 2.4e9 / mfmas, (double)wgs * 4 * mfmas * 4096 / (ms * 1e-3) / 1e12);
}

int main() {
    run<0>("V0 clumped loads, relu(a - v) right before its MFMA");
    run<1>("V1 loads spread between the MFMAs");
    run<2>("V2 relu(a - v) one k-tile ahead, loads clumped");
    run<3>("V3 relu(a - v) one k-tile ahead, everything spread");
    run<4>("V4 relu(a - v) one k-tile ahead as ONE clump with the loads, then 32 MFMAs");
    run<5>("V5 no relu at all (a as loaded), loads clumped");
    return 0;
}
