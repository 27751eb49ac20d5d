// three_nn_grid.hip — the three nearest known points through a per-frame hash grid, BIT-EXACT with the sequential scan of
// three_nn_kernel_fast (jmodt/ops/pointnet2/src/interpolate_gpu.cu:9-52): the three smallest (d2, index) pairs in
// lexicographic order (the scan's strict `<` keeps the earlier index on equal distances), float d2, best slots that were
// never filled stay at (+inf, 0) (m < 3 never comes here).
//
// The brute-force kernel (pointnet2_gather.hip) evaluates all n*m pairs (537 M at the last feature-propagation level).  Here
// the KNOWN points are binned by bq_grid_build_kernel with a cell edge chosen from their own bounding box (1.5 x their
// spacing), and every unknown point (one LANE each) walks the cells its reach touches:
//   ring 1: reach rho = h.  All points within rho of the query on every axis are in the visited cells (monotone cell
//           function, padded reach: same argument as ball_query_grid.hip), so if the third best d2 so far is below
//           rho^2 (1 - 2^-10) no unseen point — each is farther than rho on some axis — can displace or tie it: done.
//   ring 2: rho = 2 h for the lanes that could not stop.
//   rest  : queries that still cannot stop (isolated points) are appended to a list; a second kernel scans ALL known
//           points for them, one wave per query.
// Two wave-cooperative forms were built, verified bit-exact and measured SLOWER than this lane walk (round 3, kept out):
//   * one wave per bucket of unknown points, the shared neighbourhood's candidates through wave-uniform scalar loads:
//     0.54-0.59 ms (two dependent s_load round trips per cell x 27 cells per group);
//   * the same with the candidates staged 64 per load into a per-wave LDS list, coarse 2 x 2 x 2 grouping and the brute-force
//     scan's "can anybody use this candidate" filter in front of the insertion: 0.23 ms at 16384 x 4096 (scan: 0.095),
//     0.67 ms at 65536 x 4096 (this walk: 0.27, scan: 0.59) — ~13 useful lanes per fine-cell group, or ~500 candidates per
//     coarse group, cost more than they save at these sizes.
// Candidates arrive in cell order, not index order, and a bucket may be visited twice (two cells hashing to one bucket, ring
// 2 re-walking ring 1): the insertion compares (d2, index) lexicographically and ignores a point that is already in the list,
// which makes the result independent of visiting order and multiplicity — i.e. equal to the sequential scan's.
#include "jm_grid.h"

namespace jm {

constexpr int TG_MIN_M = 1024;           // known points below this: the brute-force scan is cheap
constexpr int TG_MAX_M = 16384;          // the build kernel keeps the points in registers (auto cell size needs them)

struct Top3L {
    float b1, b2, b3;
    int i1, i2, i3;
    // lexicographic (d, k) strict-less insertion, idempotent for a point already present
    __device__ __forceinline__ void visit(int k, float d) {
        const bool dup = (k == i1 && d == b1) || (k == i2 && d == b2) || (k == i3 && d == b3);
        const bool c1 = !dup && (d < b1 || (d == b1 && k < i1));
        const bool c2 = !dup && (d < b2 || (d == b2 && k < i2));
        const bool c3 = !dup && (d < b3 || (d == b3 && k < i3));
        b3 = c2 ? b2 : (c3 ? d : b3);  i3 = c2 ? i2 : (c3 ? k : i3);
        b2 = c1 ? b1 : (c2 ? d : b2);  i2 = c1 ? i1 : (c2 ? k : i2);
        b1 = c1 ? d : b1;              i1 = c1 ? k : i1;
    }
};

struct TgParams {
    int n, m, b, T;
    const float* unknown;
    const uint2* tbl;
    const float4* sorted;
    const float4* hdr;
    float* dist2;
    int* idx;
    int* todo;               // [0] = count, [1..] = b * n + point
};

__global__ void __launch_bounds__(256)
tnn_grid_query_kernel(TgParams p) {
    // whole frames per XCD: a frame's table and sorted points stay in one L2
    const int gx = (p.n + 255) / 256;
    const int slot = blockIdx.x >> 3;
    const int bi = (blockIdx.x & 7) + 8 * (slot / gx);
    if (bi >= p.b) return;
    const int pi = (slot % gx) * 256 + threadIdx.x;
    if (pi >= p.n) return;
    const float* u = p.unknown + ((size_t)bi * p.n + pi) * 3;
    const float ux = u[0], uy = u[1], uz = u[2];
    const float4 hd = p.hdr[bi];
    const float inv_h = hd.x, h = hd.y;
    const uint2* tb = p.tbl + (size_t)bi * p.T;
    const float4* so = p.sorted + (size_t)bi * p.m;
    const unsigned tmask = (unsigned)p.T - 1u;
    Top3L t{INFINITY, INFINITY, INFINITY, 0, 0, 0};
    bool done = false;
    for (int ring = 1; ring <= 2 && !done; ++ring) {
        const float rho = h * (float)ring;
        const float rpx = rho * 1.0009765625f + 3.8146973e-6f * (fabsf(ux) + 1.f);
        const float rpy = rho * 1.0009765625f + 3.8146973e-6f * (fabsf(uy) + 1.f);
        const float rpz = rho * 1.0009765625f + 3.8146973e-6f * (fabsf(uz) + 1.f);
        const int x0 = bg_cell(ux - rpx, inv_h), x1 = bg_cell(ux + rpx, inv_h);
        const int y0 = bg_cell(uy - rpy, inv_h), y1 = bg_cell(uy + rpy, inv_h);
        const int z0 = bg_cell(uz - rpz, inv_h), z1 = bg_cell(uz + rpz, inv_h);
        const long long cells = ((long long)x1 - x0 + 1) * ((long long)y1 - y0 + 1) * ((long long)z1 - z0 + 1);
        if (!(cells >= 1 && cells <= 343)) break;           // NaN / far-out coordinates: the full scan takes this query
        for (int ix = x0; ix <= x1; ++ix)
            for (int iy = y0; iy <= y1; ++iy)
                for (int iz = z0; iz <= z1; ++iz) {
                    const uint2 se = tb[bg_bucket(ix, iy, iz, tmask)];
                    for (unsigned a = se.x; a < se.y; ++a) {
                        const float4 q = so[a];
                        t.visit(__float_as_int(q.w), sqdist3(ux - q.x, uy - q.y, uz - q.z));      // (unknown - known), interpolate_gpu.cu:33
                    }
                }
        done = t.b3 < rho * rho * 0.9990234375f;
    }
    if (done) {
        float* d = p.dist2 + ((size_t)bi * p.n + pi) * 3;
        int* o = p.idx + ((size_t)bi * p.n + pi) * 3;
        d[0] = t.b1; d[1] = t.b2; d[2] = t.b3;
        o[0] = t.i1; o[1] = t.i2; o[2] = t.i3;
    } else {
        const int at = atomicAdd(p.todo, 1);
        p.todo[1 + at] = bi * p.n + pi;
    }
}

// the queries the grid walk could not finish: one wave per query scans all m known points (any order: the insertion is
// order independent), then the 64 partial lists are merged by lane 0
__global__ void __launch_bounds__(256)
tnn_grid_rest_kernel(TgParams p) {
    __shared__ float sd[4][3][64];
    __shared__ int si[4][3][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int total = p.todo[0];
    for (int w = blockIdx.x * 4 + wave; w < total; w += gridDim.x * 4) {
        const int q = p.todo[1 + w];
        const int bi = q / p.n;
        const float* u = p.unknown + (size_t)q * 3;
        const float ux = u[0], uy = u[1], uz = u[2];
        const float4* so = p.sorted + (size_t)bi * p.m;
        Top3L t{INFINITY, INFINITY, INFINITY, 0, 0, 0};
        for (int a = lane; a < p.m; a += 64) {
            const float4 v = so[a];
            t.visit(__float_as_int(v.w), sqdist3(ux - v.x, uy - v.y, uz - v.z));
        }
        sd[wave][0][lane] = t.b1; sd[wave][1][lane] = t.b2; sd[wave][2][lane] = t.b3;
        si[wave][0][lane] = t.i1; si[wave][1][lane] = t.i2; si[wave][2][lane] = t.i3;
        __threadfence_block();
        if (lane == 0) {
            Top3L r{INFINITY, INFINITY, INFINITY, 0, 0, 0};
            for (int l = 0; l < 64; ++l)
                for (int s = 0; s < 3; ++s)
                    if (sd[wave][s][l] < INFINITY) r.visit(si[wave][s][l], sd[wave][s][l]);
            float* d = p.dist2 + (size_t)q * 3;
            int* o = p.idx + (size_t)q * 3;
            d[0] = r.b1; d[1] = r.b2; d[2] = r.b3;
            o[0] = r.i1; o[1] = r.i2; o[2] = r.i3;
        }
        __threadfence_block();
    }
}

static bool tg_applies(int n, int m) { return m >= TG_MIN_M && m <= TG_MAX_M && n >= 1; }

}  // namespace jm

using namespace jm;

/* policy: where the grid walk beats the scan.  One lane per unknown point walks ~27 cells x a few known points each, a chain
 * of dependent L2 round trips (~0.2-0.3 ms whatever the size, measured), while the scan streams the known set through the scalar
 * cache at full VALU rate: the walk wins from ~1.3e8 pairs on (65536 x 4096: 0.27 vs 0.59 ms; 16384 x 4096: 0.32 vs 0.095 ms) */
extern "C" size_t jm_three_nn_workspace_bytes(int b, int n, int m) {
    if ((long long)n * m < (1LL << 27)) return 0;
    return jm_three_nn_grid_workspace_bytes(b, n, m);
}

/* capability: the workspace with which jm_three_nn_ws takes the grid walk for ANY n (1024 <= m <= 16384) */
extern "C" size_t jm_three_nn_grid_workspace_bytes(int b, int n, int m) {
    if (b < 1 || !tg_applies(n, m)) return 0;
    return align_up((size_t)b * BG_T_MAX * sizeof(uint2), 256) + align_up((size_t)b * m * sizeof(float4), 256) +
           align_up((size_t)b * sizeof(float4), 256) + align_up(((size_t)b * n + 1) * sizeof(int), 256);
}

extern "C" int jm_three_nn_ws(int b, int n, int m, const float* unknown, const float* known, float* dist2, int* idx, void* ws,
                              size_t ws_bytes, jm_stream_t stream) {
    const size_t need = jm_three_nn_grid_workspace_bytes(b, n, m);
    if (!ws || need == 0) return jm_three_nn(b, n, m, unknown, known, dist2, idx, stream);
    JM_REQUIRE(unknown && known && dist2 && idx, "three_nn: null pointer");
    JM_REQUIRE((long long)b * n < (1LL << 31), "three_nn: too many points");
    if (ws_bytes < need) { set_error("three_nn: workspace %zu < %zu bytes", ws_bytes, need); return JM_EWORKSPACE; }
    JM_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 15u) == 0, "three_nn: workspace must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    char* w = (char*)ws;
    uint2* tbl = (uint2*)w;      w += align_up((size_t)b * BG_T_MAX * sizeof(uint2), 256);
    float4* sorted = (float4*)w; w += align_up((size_t)b * m * sizeof(float4), 256);
    float4* hdr = (float4*)w;    w += align_up((size_t)b * sizeof(float4), 256);
    int* todo = (int*)w;
    (void)jm_zero_async(todo, sizeof(int), s);
    int T = 4096;
    while (T < m && T < BG_T_MAX) T *= 2;
#define JM_TG_BUILD(TPT)                                                                                                       \
    do {                                                                                                                      \
        (void)hipFuncSetAttribute((const void*)bq_grid_build_kernel<TPT, 16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                  1024 * TPT * 4);                                                                            \
        hipLaunchKernelGGL((bq_grid_build_kernel<TPT, 16, true>), dim3((unsigned)b), dim3(1024), 1024 * TPT * 4, s, m, 0.f,    \
                           known, tbl, sorted, hdr);                                                                          \
    } while (0)
    if (T == 4096) JM_TG_BUILD(4);
    else if (T == 8192) JM_TG_BUILD(8);
    else JM_TG_BUILD(16);
#undef JM_TG_BUILD
    TgParams p{};
    p.n = n; p.m = m; p.b = b; p.T = T;
    p.unknown = unknown; p.tbl = tbl; p.sorted = sorted; p.hdr = hdr; p.dist2 = dist2; p.idx = idx; p.todo = todo;
    const long long groups = (long long)divup(b, 8) * 8 * divup(n, 256);
    JM_REQUIRE(groups < (1LL << 31), "three_nn: too many points");
    hipLaunchKernelGGL(tnn_grid_query_kernel, dim3((unsigned)groups), dim3(256), 0, s, p);
    hipLaunchKernelGGL(tnn_grid_rest_kernel, dim3(512), dim3(256), 0, s, p);
    return check_launch("three_nn(grid)");
}
