// fps_common.h — pieces shared by fps.hip (the kernels in use) and fps_pruned.hip (the exact spatially
// pruned variants kept for reference: bit-exact, measured no faster; selected with JM_FPS_PRUNE).
#pragma once
#include "jm_common.h"

namespace jm {

typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned bitrev_u(unsigned v, int bits) { return __brev(v) >> (32 - bits); }

constexpr int FPS_OUT_CHUNK = 4096;

// variant 1: wave clusters, variant 2: slot clusters; false when the shape is not covered (n must be 4096 /
// 8192 / 16384 with the reference's 1024-thread block)
bool launch_fps_pruned(int variant, int b, int n, int m, const float* xyz, float* temp, int* idx, hipStream_t s);

}  // namespace jm
