// tools/mfma_v2_probe.hip — would a different tile geometry lift the fused SA kernel?  Bare model of the proposed inner
// structure: 8 MFMA waves per workgroup (two per SIMD), each a 32-point x 64-channel block; activations ROW-major in LDS
// ([point][K + 4]), read as two ds_read_b128 per k-tile (k = 16 kt + 8 lk + kk); hoisted layer: relu(a - v) with v from an
// LDS table; hidden layer computed TRANSPOSED (A = weights, B = activations) so that its epilogue is 4 ds_write_b128 per
// accumulator; last layer in the normal orientation (rows = points in registers: cheap max-pool).  Weights from global.
//   hipcc --offload-arch=gfx950 -O3 -w tools/mfma_v2_probe.hip -o tools/bin/mfma_v2_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define SB() __builtin_amdgcn_sched_barrier(0)
constexpr int S = 132;     // row stride (floats): S / 4 odd -> conflict-free b128 reads of 16 rows

template <int MODE>        // bit 0: relu(a - v) on the first layer; bit 1: layer boundaries; bit 2: loads spread between the MFMAs
__global__ void __launch_bounds__(512) probe(const float* __restrict__ W, float* out, int tiles) {
    __shared__ float X[2][128 * S];
    __shared__ float VT[4 * 128];
    for (int i = threadIdx.x; i < 2 * 128 * S; i += 512) (&X[0][0])[i] = (i % 977) * 1e-3f;
    for (int i = threadIdx.x; i < 512; i += 512) VT[i] = (i % 13) * 0.05f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 31, lk = lane >> 5;
    const int rb = wave >> 1, cb = wave & 1;              // 4 row blocks x 2 column halves (64 channels each)
    const float* bp = W + ((size_t)cb * 64 + lr) * 16 + lk * 8;     // packed [kt][n][lk][8]
    f32x16 acc[2];
    float sink = 0.f;
    for (int tile = 0; tile < tiles; ++tile) {
        for (int layer = 0; layer < 2; ++layer) {
            const float* A = &X[layer][0] + (size_t)(rb * 32 + lr) * S + lk * 8;
            const float* vt = VT + rb * 128 + lk * 8;
            for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.5f;
            float4 a0[2], a1[2], v0[2], v1[2], b0[4], b1[4];
            auto loadA = [&](float4 (&a)[2], float4 (&v)[2], int kt) __attribute__((always_inline)) {
                a[0] = *reinterpret_cast<const float4*>(A + kt * 16); a[1] = *reinterpret_cast<const float4*>(A + kt * 16 + 4);
                if ((MODE & 1) && layer == 0) { v[0] = *reinterpret_cast<const float4*>(vt + kt * 16); v[1] = *reinterpret_cast<const float4*>(vt + kt * 16 + 4); }
            };
            auto loadB = [&](float4 (&b)[4], int kt) __attribute__((always_inline)) {
                const float* q = bp + (size_t)(kt & 7) * 2048;
                b[0] = *reinterpret_cast<const float4*>(q); b[1] = *reinterpret_cast<const float4*>(q + 4);
                b[2] = *reinterpret_cast<const float4*>(q + 512); b[3] = *reinterpret_cast<const float4*>(q + 516);
            };
            auto mm = [&](const float4 (&a)[2], const float4 (&v)[2], const float4 (&b)[4]) __attribute__((always_inline)) {
                float t[8] = {a[0].x, a[0].y, a[0].z, a[0].w, a[1].x, a[1].y, a[1].z, a[1].w};
                if ((MODE & 1) && layer == 0) {
                    const float w[8] = {v[0].x, v[0].y, v[0].z, v[0].w, v[1].x, v[1].y, v[1].z, v[1].w};
#pragma unroll
                    for (int kk = 0; kk < 8; ++kk) t[kk] = fmaxf(t[kk] - w[kk], 0.f);
                }
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    const float w0 = reinterpret_cast<const float*>(&b[0])[kk], w1 = reinterpret_cast<const float*>(&b[2])[kk];
                    if (layer == 0) {      // transposed: A = weights, B = activations
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0, t[kk], acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1, t[kk], acc[1], 0, 0, 0);
                    } else {
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(t[kk], w0, acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(t[kk], w1, acc[1], 0, 0, 0);
                    }
                }
            };
            loadA(a0, v0, 0); loadB(b0, 0);
            auto spread = [&]() __attribute__((always_inline)) {     // 16 MFMAs, one load and one or two VALU after each of the first 8
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (i < 4) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    else if (i < 8) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
                }
            };
            for (int kt = 0; kt < 8; kt += 2) {
                loadA(a1, v1, kt + 1); loadB(b1, kt + 1);
                if (!(MODE & 4)) SB();
                mm(a0, v0, b0);
                if (MODE & 4) spread();
                SB();
                loadA(a0, v0, (kt + 2) & 7); loadB(b0, kt + 2);
                if (!(MODE & 4)) SB();
                mm(a1, v1, b1);
                if (MODE & 4) spread();
                SB();
            }
            if (MODE & 2) {
                if (layer == 0) {          // hidden epilogue: relu -> row-major tile of the next layer, 4 b128 per accumulator
                    float* Y = &X[1][0] + (size_t)(rb * 32 + lr) * S + cb * 64 + 4 * lk;
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int rq = 0; rq < 4; ++rq) {
                            float4 v;
                            v.x = fmaxf(acc[j][4 * rq + 0], 0.f); v.y = fmaxf(acc[j][4 * rq + 1], 0.f);
                            v.z = fmaxf(acc[j][4 * rq + 2], 0.f); v.w = fmaxf(acc[j][4 * rq + 3], 0.f);
                            *reinterpret_cast<float4*>(Y + j * 32 + 8 * rq) = v;
                        }
                } else {                   // max-pool over the block's 32 points: in-register over 16, then across lk
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        float m = acc[j][0];
#pragma unroll
                        for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[j][r]);
                        m = fmaxf(m, __shfl_xor(m, 32));
                        sink += m;
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            } else {
                for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) sink += acc[j][r];
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = sink;
}

template <int MODE>
void run(const char* what) {
    const int wgs = 256, tiles = 400;
    float *out, *W;
    hipMalloc(&out, sizeof(float) * wgs * 512);
    hipMalloc(&W, sizeof(float) * 2048 * 16);
    std::vector<float> h(2048 * 16);
    for (size_t i = 0; i < h.size(); ++i) h[i] = ((i * 2654435761u) % 1000) * 1e-4f - 0.05f;
    hipMemcpy(W, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe<MODE>, dim3(wgs), dim3(512), 0, 0, W, out, 4);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<MODE>, dim3(wgs), dim3(512), 0, 0, W, out, tiles);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)wgs * 8 * tiles * 2 * 8 * 16;      // per wave per tile: 2 layers x 8 k-tiles x 16
    printf("%-64s %.3f ms  %.1f TF  (%.1f cycles @2.4GHz per MFMA per SIMD)\n", what, ms, mfmas * 4096 / (ms * 1e-3) / 1e12,
           ms * 1e-3 * 2.4e9 / ((double)tiles * 2 * 8 * 16 * 2));
}

int main() {
    run<0>("8 waves, row-major b128 feeds, no VALU, no boundaries");
    run<1>("+ relu(a - v) on the first layer");
    run<2>("+ layer boundaries (b128 epilogue / max-pool + barrier)");
    run<3>("+ both (= the proposed kernel's MFMA side)");
    run<7>("+ both, loads / VALU spread between the MFMAs (sched_group_barrier)");
    return 0;
}
