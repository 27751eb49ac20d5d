// sort.hip — stable descending argsort of up to 16384 floats per row, one workgroup per row, entirely in LDS (gfx950).
//
// proposal_layer.py:45 (`torch.sort(scores, descending=True)` over the 16384 point scores of a frame, in front of the
// distance-band split and the NMS walk): as a library call it is a 14-kernel merge sort, 0.22 ms per batch of 8 frames on the
// main chain's critical path.  Here: LSD radix sort, 8 passes of 4 bits, 1024 threads x 16 elements held in REGISTERS between
// passes (striped: element e of thread t is position e * 1024 + t, so every LDS access is conflict-free):
//   count   : a wave finds the lanes that share its element's digit with four ballots (rank = popcount below the lane), the
//             first lane of each digit group writes the group size to hist[digit][e][wave]  (4096 counters, position order);
//   scan    : exclusive scan of the 4096 counters (4 per thread, wave scan by DPP-free shuffles, 16 wave totals through LDS);
//   scatter : key and 16-bit index to LDS at hist[...] + rank; read back in position order.
// Stable by construction (groups are laid out in position order, ranks keep lane order), so equal scores keep their index
// order = torch.sort(stable=True); keys: float bits mapped to an order-preserving unsigned, inverted for "descending";
// -0.0 is ordered as +0.0; NaNs order by their bit pattern (+NaN above +inf, -NaN below -inf), as the library's radix sort does.
#include "jm_common.h"

namespace jm {

constexpr int SRT_T = 1024, SRT_MAXE = 16, SRT_MAXN = SRT_T * SRT_MAXE;

__device__ __forceinline__ unsigned srt_key_desc(float f) {
    unsigned u = __float_as_uint(f);
    if ((u & 0x7FFFFFFFu) == 0u) u = 0u;                               // -0.0 == +0.0
    const unsigned asc = u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);  // ascending order of the floats
    return ~asc;                                                       // descending
}

template <int E>
__global__ void __launch_bounds__(SRT_T)
argsort_desc_kernel(int n, const float* __restrict__ scores, long long* __restrict__ order) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];
    unsigned* K = lds;                                                  // [E * 1024] keys
    unsigned* hist = K + E * SRT_T;                                     // [16 digits][E][16 waves]
    unsigned short* I = reinterpret_cast<unsigned short*>(hist + 16 * E * 16);   // [E * 1024] indices
    unsigned* wtot = reinterpret_cast<unsigned*>(I + E * SRT_T);       // [16] wave totals of the scan
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* row = scores + (size_t)blockIdx.x * n;
    unsigned key[E];
    unsigned short idx[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int p = e * SRT_T + tid;
        key[e] = p < n ? srt_key_desc(row[p]) : 0xFFFFFFFFu;            // padding sorts last
        idx[e] = (unsigned short)p;
    }
    const unsigned long long lt = (1ull << lane) - 1ull;
    constexpr int NH = 16 * E * 16;                                     // counters
    for (int shift = 0; shift < 32; shift += 4) {
        for (int i = tid; i < NH; i += SRT_T) hist[i] = 0u;
        __syncthreads();
        unsigned rank[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const unsigned d = (key[e] >> shift) & 15u;
            unsigned long long m = ~0ull;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned long long b = __ballot((d >> k) & 1u);
                m &= ((d >> k) & 1u) ? b : ~b;
            }
            rank[e] = (unsigned)__popcll(m & lt);
            if (rank[e] == 0u) hist[(d * E + e) * 16 + wave] = (unsigned)__popcll(m);
        }
        __syncthreads();
        // exclusive scan of hist[0 .. NH): NH / 1024 = E / 4... counters per thread (E = 16: 4; smaller E: strided single pass below)
        {
            constexpr int PER = NH / SRT_T > 0 ? NH / SRT_T : 1;        // E >= 4: E / 4 consecutive counters per thread
            const bool act = tid * PER < NH;
            unsigned v[PER], s = 0;
#pragma unroll
            for (int j = 0; j < PER; ++j) { v[j] = act ? hist[tid * PER + j] : 0u; s += v[j]; }
            unsigned inc = s;                                           // inclusive scan over the wave
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned t = __shfl_up(inc, o);
                if (lane >= o) inc += t;
            }
            if (lane == 63) wtot[wave] = inc;
            __syncthreads();
            unsigned base = 0;
            for (int w = 0; w < wave; ++w) base += wtot[w];
            unsigned run = base + inc - s;
#pragma unroll
            for (int j = 0; j < PER; ++j) { if (act) hist[tid * PER + j] = run; run += v[j]; }
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const unsigned d = (key[e] >> shift) & 15u;
            const unsigned dst = hist[(d * E + e) * 16 + wave] + rank[e];
            K[dst] = key[e];
            I[dst] = idx[e];
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < E; ++e) { key[e] = K[e * SRT_T + tid]; idx[e] = I[e * SRT_T + tid]; }
        // (the next pass writes K / I only after three more barriers)
    }
    long long* out = order + (size_t)blockIdx.x * n;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int p = e * SRT_T + tid;
        if (p < n) out[p] = (long long)idx[e];
    }
}

template <int E>
static void launch_argsort(int b, int n, const float* scores, long long* order, hipStream_t s) {
    const size_t lds = (size_t)E * SRT_T * 4 + (size_t)16 * E * 16 * 4 + (size_t)E * SRT_T * 2 + 64;
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(argsort_desc_kernel<E>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((argsort_desc_kernel<E>), dim3((unsigned)b), dim3(SRT_T), lds, s, n, scores, order);
}

}  // namespace jm

using namespace jm;

extern "C" int jm_argsort_desc_supported(int n) { return n >= 1 && n <= SRT_MAXN; }

extern "C" int jm_argsort_desc_stable(int b, int n, const float* scores, long long* order, jm_stream_t stream) {
    JM_REQUIRE(b >= 0 && n >= 0, "argsort_desc: negative size");
    if (b == 0 || n == 0) return JM_OK;
    JM_REQUIRE(n <= SRT_MAXN, "argsort_desc: at most 16384 elements per row");
    JM_REQUIRE(scores && order, "argsort_desc: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const int e = divup(n, SRT_T);
    if (e <= 4) launch_argsort<4>(b, n, scores, order, s);
    else if (e <= 8) launch_argsort<8>(b, n, scores, order, s);
    else launch_argsort<16>(b, n, scores, order, s);
    return check_launch("argsort_desc");
}
