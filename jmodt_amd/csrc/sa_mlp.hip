// sa_mlp.hip — fused set-abstraction block on the fp32 matrix cores (gfx950):
//     group (xyz - centre | features)  ->  [1x1 conv + BN(eval, folded) + ReLU] x L  ->  max over nsample
//
// Replaces the per-scale body of _PointnetSAModuleBase.forward
// (jmodt/ops/pointnet2/pointnet2_modules.py:46-52): QueryAndGroup's two grouping_operation launches,
// the subtraction, the cat (pointnet2_utils.py:249-264), SharedMLP's cuDNN 1x1 convolutions on the
// materialised (B, 3+C, npoint, nsample) tensor (pytorch_utils.py:6-33) and F.max_pool2d.
// For the RCNN's first SA level that grouped tensor alone is 4.4 GB per 1024 RoIs;
// here neither it nor any hidden activation ever leaves the chip.
//
// Persistent workgroups (one per CU), each walking over tiles of 128 consecutive (centre, sample) rows,
// split into two roles (wave specialisation):
//   * waves 0-3, one per SIMD, do nothing but MFMA + epilogues.  Activations live in two
//     144(k) x 128(row) LDS buffers, k-major, used in ping-pong: a layer reads its A operand straight
//     from one and its epilogue (bias + ReLU) writes the other with 16-byte stores in the accumulator's
//     own row order.  Weights never touch LDS: they are pre-packed (jm_sa_mlp_pack) so that the B
//     operand of lane (col, khalf) for one 16-deep k-tile is 8 consecutive floats -> two
//     global_load_dwordx4 per 32-column block per k-tile, coalesced over the wave, served by L1/L2
//     (a layer is <= 64 KB and every workgroup reads the same one), prefetched one k-tile ahead in
//     registers; the last prefetch of a stage fetches the NEXT stage's first operand, so no stage
//     starts on a cold load and the k-loops contain no barrier.
//   * waves 4-7 gather: layer 1's input (first 3 channels = xyz - centre, then the point features
//     through idx) in chunks of 144 channels — chunk c+1 while chunk c is being multiplied, and the next
//     tile's first chunk during the current tile's later layers (its loads are in flight for two layers
//     before they are parked in the buffer that has just become free).  The gather's long scattered-load
//     latencies and its vmcnt waits therefore never touch the MFMA waves.
//   Both roles execute the same sequence of workgroup barriers (one per stage).
//   * Whole frames per XCD when there are enough of them: a frame's features are then pulled into one
//     L2 instead of all eight.
//   * Last layer: any width (128-column tiles); its epilogue reduces max over each centre's nsample
//     rows straight out of the accumulator layout, then bias + ReLU (both commute with max).
// v_mfma_f32_32x32x2_f32 everywhere: exact-f32 products (1e-4 parity with the fp32 reference path).
// Constraints (the Python module falls back to the unfused path otherwise): nsample in {16,32,64},
// npoint*nsample % 128 == 0, hidden widths <= 128, eval-mode BatchNorm (folded by the caller).
#include "jm_mfma.h"

namespace jm {

constexpr int SM_KC = 144;   // channels per activation buffer: 3 + 128 (both RCNN SA levels) in one chunk
constexpr int SM_GRP = SM_KC / 16;                     // gather groups (16 channels each) per chunk
constexpr int SM_BUF = SM_KC * SM_LDP;                 // floats per activation buffer
constexpr size_t SM_LDS_BYTES = 2 * (size_t)SM_BUF * sizeof(float);
constexpr int SM_VT = 8 * 128;                         // pre-projected mode: per-centre table, <= 8 centres x <= 128 channels
constexpr size_t SM_LDS_BYTES_PRE = SM_LDS_BYTES + SM_VT * sizeof(float);

struct SaMlpParams {
    int N, M, C, ns;                 // points per frame, centres per frame, feature channels, nsample
    const float* xyz;                // (B,N,3)
    const float* new_xyz;            // (B,M,3)
    const float* feat;               // (B,C,N) or null
    const int* idx;                  // (B,M,ns)
    const float* w1x;                // (C,4) or null.  Non-null = "pre-projected" mode: `feat` holds u = W1 [xyz | f] + b1
                                     // per POINT and w1x the xyz columns of W1 (4th column padding); the gather forms
                                     // relu(u[idx] - W1x . centre) = the first layer's output, and the layers given are 2..L
    int L;                           // layers (1..4)
    int kp[5];                       // kp[l] = pad16(width_l), l = 0..L
    int np[4];                       // np[l] = pad128(width_{l+1}): packed rows of layer l
    const float* W[4];               // packed weights of layer l (see jm_sa_mlp_pack)
    const float* bias[4];            // (np[l]) zero padded
    float* out;                      // (B, cout, M); frame b starts at out + b * obs (obs = cout * M unless the caller writes into
    int cout;                        // a channel slice of a wider (B, Ctot, M) tensor: an MSG module's concatenation)
    size_t obs;
    int tiles_per_frame, total_tiles, xcd_frames;
#ifdef JM_TOOLS_BUILD
    long long* trace;                // tools build: shader-clock stamps of workgroup 0's first MFMA wave (tools/sa_trace.py)
    int dbg;                         // tools build (JM_SA_DBG): 2 = gather role idles (timing experiment, wrong results)
#endif
};

#ifdef JM_TOOLS_BUILD
#define JM_SA_STAMP(k) do { if (p.trace && blockIdx.x == 0 && tid == 0 && it < 64) p.trace[it * 8 + (k)] = (long long)clock64(); } while (0)
#else
#define JM_SA_STAMP(k) do { } while (0)
#endif

// packed layout: Wp[kt][n][khalf][kk] = W'[n][16 kt + 2 kk + khalf]   (kt < Kp/16, n < Np, khalf < 2, kk < 8)
// W' = W zero padded, except for the FIRST layer, whose input channels [xyz(3) | C features]
// (QueryAndGroup's cat order, pointnet2_utils.py:258-262) are reordered to
//     [C features | zeros to pad16(C) | xyz(3) | zeros to +16]
// so that the gather works in whole 16-channel groups of plain feature rows, with xyz in a group of its own.

__global__ void sa_mlp_pack_kernel(int cout, int cin, int Kp, int Np, int first, const float* __restrict__ w,
                                   const float* __restrict__ b, float* __restrict__ wp, float* __restrict__ bp) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < Np) bp[e] = (e < cout && b) ? b[e] : 0.f;
    if (e < sa_kmajor_elems(cout, cin)) wp[(size_t)Kp * Np + e] = w[(size_t)(e % cout) * cin + e / cout];   // [k][n]: k = e / cout
    if (e >= Kp * Np) return;
    const int kk = e & 7, kh = (e >> 3) & 1, n = (e >> 4) % Np, kt = (e >> 4) / Np;
    const int k = 16 * kt + 2 * kk + kh;
    int src = k;                                  // column of the unpacked weight, or -1 for padding
    if (first) {
        const int C = cin - 3, Cp = pad_to(C, 16);
        src = k < C ? 3 + k : ((k >= Cp && k < Cp + 3) ? k - Cp : -1);
    } else if (k >= cin) {
        src = -1;
    }
    wp[e] = (n < cout && src >= 0) ? w[(size_t)n * cin + src] : 0.f;
}

// ---- persistent tile schedule, shared by both roles.  xcd_frames: workgroup b runs on XCD b % 8
// (round-robin dispatch), so every XCD gets whole frames; otherwise tiles are dealt out linearly.
struct SaSchedule {
    int n_local;
    int xcd, slot, per, nwg, tpf, mode;
    __device__ SaSchedule(const SaMlpParams& p) {
        nwg = gridDim.x; tpf = p.tiles_per_frame; mode = p.xcd_frames;
        xcd = blockIdx.x & 7; slot = blockIdx.x >> 3; per = nwg >> 3;
        if (mode) {
            const int nb = p.total_tiles / tpf;
            const int local_tiles = ((nb - xcd + 7) >> 3) * tpf;
            n_local = local_tiles > slot ? (local_tiles - slot + per - 1) / per : 0;
        } else {
            n_local = p.total_tiles > (int)blockIdx.x ? (p.total_tiles - (int)blockIdx.x + nwg - 1) / nwg : 0;
        }
    }
    __device__ void tile(int i, int& bi, int& row0) const {
        if (mode) {
            const int lt = slot + i * per;
            bi = xcd + 8 * (lt / tpf);
            row0 = (lt % tpf) * SM_BM;
        } else {
            const int t = blockIdx.x + i * nwg;
            bi = t / tpf;
            row0 = (t % tpf) * SM_BM;
        }
    }
};

// =============================================================================== gather role
// 256 threads: thread fills row `grow` of the tile, channels gk0, gk0 + 2, ... of a chunk.  gk0 is
// wave-uniform: with it in an SGPR every channel index, base pointer and padding test is scalar and the
// gathers use the saddr + 32-bit-offset form (no per-element address VGPRs).
struct SaRow {
    int gidx;            // this thread's neighbour index
    float cx, cy, cz;    // its centre (named fields: a runtime-indexed array would live in scratch)
    int bi, centre;
};

__device__ __forceinline__ void sa_gather_role(const SaMlpParams& p, float* lds, int ltid) {
    const SaSchedule sch(p);
    if (sch.n_local == 0) return;
    const int grow = ltid & 127, gk0 = __builtin_amdgcn_readfirstlane(ltid >> 7);
    const int K0 = p.kp[0];
    const int nchunks = (K0 + SM_KC - 1) / SM_KC;
    const bool single = (p.L == 1);
    // first-layer channel order (see sa_mlp_pack_kernel): nfast whole groups of features, an optional
    // partial feature group (index nfast), then the xyz group (index gxyz = pad16(C)/16, the last one)
    const bool pre = p.w1x != nullptr;
    const int C = p.C, nfast = C >> 4, gxyz = pre ? (1 << 20) : (C + 15) >> 4;   // pre-projected mode has no xyz group
    const bool has_tail = (C & 15) != 0;

    auto load_row = [&](int i) {
        SaRow t;
        int row0;
        sch.tile(i, t.bi, row0);
        t.gidx = p.idx[(size_t)t.bi * p.M * p.ns + row0 + grow];
        t.centre = (row0 + grow) / p.ns;
        const float* cp = p.new_xyz + ((size_t)t.bi * p.M + t.centre) * 3;
        t.cx = cp[0]; t.cy = cp[1]; t.cz = cp[2];
        return t;
    };
    float g[8 * SM_GRP];   // one chunk in flight: SM_KC channels x 128 rows / 256 threads
    float gt[8], gx[2];    // the partial feature group and the xyz group of this thread
    // A whole feature group is the lean path: one scalar base pointer, 8 loads / 8 ds_writes and nothing per
    // element in between.  This matters: the wave shares its SIMD with an MFMA wave and gets few issue slots.
    auto issue = [&](const SaRow& t, int c) {
#ifdef JM_TOOLS_BUILD
        if (p.dbg == 2) return;
#endif
        const float* xyz_b = p.xyz + (size_t)t.bi * p.N * 3;
        const float* feat_b = p.feat ? p.feat + (size_t)t.bi * C * p.N : xyz_b;   // never dereferenced when C == 0
        const unsigned off_f = (unsigned)t.gidx;
        const size_t two_n = 2 * (size_t)p.N;
        const int g0 = c * SM_GRP;                           // first group of the chunk
        const float* cb = feat_b + ((size_t)g0 * 16 + gk0) * p.N;
#pragma unroll
        for (int grp = 0; grp < SM_GRP; ++grp) {
            if (g0 + grp < nfast) {                          // uniform
                const float* rp = cb + (size_t)grp * 16 * p.N;
#pragma unroll
                for (int j = 0; j < 8; ++j) g[grp * 8 + j] = rp[j * two_n + off_f];

                // one group's scalar base pointers at a time (left alone, the scheduler forms all of them first
                // and spills SGPRs into VGPR lanes, which costs VALU slots on the MFMA wave's SIMD)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (has_tail && nfast >= g0 && nfast < g0 + SM_GRP) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int kcl = min(nfast * 16 + gk0 + 2 * j, C - 1);      // unconditional load, clamped address
                gt[j] = feat_b[(size_t)kcl * p.N + off_f];
            }
        }
        if (gxyz >= g0 && gxyz < g0 + SM_GRP) {
            const float* q = xyz_b + (size_t)t.gidx * 3;
            gx[0] = q[gk0];                                  // x (gk0 = 0) or y (gk0 = 1)
            gx[1] = q[2];                                    // z, used by gk0 = 0 only
        }
    };
    auto store = [&](const SaRow& t, int c, float* G) {   // centre subtraction / zero padding happen here
#ifdef JM_TOOLS_BUILD
        if (p.dbg == 2) return;
#endif
        float* Gt = G + gk0 * SM_LDP + grow;                 // element i of this thread: + 2 i SM_LDP
        const int g0 = c * SM_GRP;
#pragma unroll
        for (int grp = 0; grp < SM_GRP; ++grp) {
            if (g0 + grp < nfast) {
#pragma unroll
                for (int j = 0; j < 8; ++j) Gt[2 * (grp * 8 + j) * SM_LDP] = g[grp * 8 + j];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (has_tail && nfast >= g0 && nfast < g0 + SM_GRP) {
            float* Gq = Gt + (size_t)(nfast - g0) * 16 * SM_LDP;
#pragma unroll
            for (int j = 0; j < 8; ++j) Gq[2 * j * SM_LDP] = (nfast * 16 + gk0 + 2 * j < C) ? gt[j] : 0.f;
        }
        if (gxyz >= g0 && gxyz < g0 + SM_GRP) {
            float* Gq = Gt + (size_t)(gxyz - g0) * 16 * SM_LDP;
            Gq[0] = gx[0] - (gk0 == 0 ? t.cx : t.cy);
            Gq[2 * SM_LDP] = gk0 == 0 ? gx[1] - t.cz : 0.f;
#pragma unroll
            for (int j = 2; j < 8; ++j) Gq[2 * j * SM_LDP] = 0.f;
        }
    };

    // pre-projected mode: v[c][k] = W1x[k] . centre_c for the tile's 128 / ns centres, consumed by the MFMA waves as
    // relu(u - v) on their A operand (jm_mfma.h, SUBV).  Layout [centre][k-tile][khalf][8] = what one lane reads per
    // k-tile.  ONE table: the MFMA waves read it only during the first layer of a tile, which is complete at the first
    // hidden epilogue's barrier; the table of the next tile is written after that barrier (host: >= 2 layers here).
    float* vt = lds + 2 * SM_BUF;
    auto write_table = [&](int i) {
        int bi, row0;
        sch.tile(i, bi, row0);
        const int ncen = SM_BM / p.ns, c0 = row0 / p.ns;
        for (int e = ltid; e < ncen * C; e += 256) {
            const int c = e / C, k = e - c * C;
            const float* cp = p.new_xyz + ((size_t)bi * p.M + c0 + c) * 3;
            const float* wv = p.w1x + (size_t)k * 4;
            const float val = __builtin_fmaf(wv[2], cp[2], __builtin_fmaf(wv[1], cp[1], wv[0] * cp[0]));
            vt[(size_t)c * C + (k >> 4) * 16 + (k & 1) * 8 + ((k & 15) >> 1)] = val;
        }
    };
    SaRow ct = load_row(0);
    issue(ct, 0);
    if (pre) write_table(0);
    store(ct, 0, lds);
    lds_barrier();                                                   // B0: chunk 0 of the first tile published
    int cur = 0;
    for (int it = 0; it < sch.n_local; ++it) {
        const bool has_next = it + 1 < sch.n_local;
        const SaRow nt = load_row(has_next ? it + 1 : it);
        if (!single) {
            for (int c = 0; c + 1 < nchunks; ++c) {                  // chunk c+1 while chunk c is multiplied
                issue(ct, c + 1);
                store(ct, c + 1, lds + (cur ^ 1) * SM_BUF);
                lds_barrier();
                cur ^= 1;
            }
        }
        if (single) {
            if (has_next) issue(nt, 0);
        } else {
            // one-chunk inputs (every RCNN level, and every pre-projected call): the gather registers are free for the
            // whole first layer of this tile, so the next tile's loads are issued BEFORE the first hidden epilogue's
            // barrier and have that layer's MFMA time to land; only the LDS stores are left for after the barrier (with
            // two layers per tile the loads otherwise had to complete within the last layer's MFMAs)
            const bool early = has_next && nchunks == 1;
            if (early) issue(nt, 0);
            lds_barrier(); cur ^= 1;                                 // first hidden epilogue's barrier
            if (has_next && !early) issue(nt, 0);
            if (pre && has_next) write_table(it + 1);                // the MFMA waves are past this tile's first layer
            for (int l = 1; l < p.L - 1; ++l) { lds_barrier(); cur ^= 1; }   // the other hidden epilogues' barriers
        }
        if (has_next) {
            store(nt, 0, lds + (cur ^ 1) * SM_BUF);                  // the buffer the last layer does not read
            lds_barrier();
            cur ^= 1;
        }
        ct = nt;
    }
}

// =============================================================================== MFMA role
__device__ __forceinline__ void sa_mfma_role(const SaMlpParams& p, float* lds, int tid) {
    const SaSchedule sch(p);
    if (sch.n_local == 0) return;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, lk = lane >> 5;
    const int a_off = lk * SM_LDP + wm * 64 + lr;
    const bool single = (p.L == 1);
    const int L = p.L;
    auto KP = [&](int i) { return p.kp[i]; };
    auto NP = [&](int i) { return p.np[i]; };
    auto WW = [&](int i) { return p.W[i]; };
    auto BS = [&](int i) { return p.bias[i]; };
    const int kp1 = p.kp[1];
    const float* bs0 = p.bias[0];
    const int K0 = p.kp[0];
    const int nchunks = (K0 + SM_KC - 1) / SM_KC;

    f32x16 acc[2][2];
    // accumulators start at the layer's bias (the epilogues then only clamp): columns ncol0 + 32 j + lr
    auto init_acc = [&](const float* bl, int ncol0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float bv = bl[ncol0 + j * 32 + lr];          // biases are zero padded to np (multiple of 128)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = bv;
        }
    };
    // this lane's packed-weight pointer: layer l, k-tile kt0, column block starting at ncol0
    auto wptr = [&](int l, int kt0, int ncol0) {
        return WW(l) + ((size_t)kt0 * NP(l) + ncol0 + lr) * 16 + lk * 8;
    };
    float4 bpre[4];
    auto preload = [&](const float* q) {
        bpre[0] = *reinterpret_cast<const float4*>(q);
        bpre[1] = *reinterpret_cast<const float4*>(q + 4);
        bpre[2] = *reinterpret_cast<const float4*>(q + 512);
        bpre[3] = *reinterpret_cast<const float4*>(q + 516);
    };
    // ncols: valid (padded-to-16) columns from ncol0 on; a wave without columns idles but keeps the
    // preload chain going
    // pre-projected mode: this lane's rows of the per-centre table (gather role: write_table) for its two 32-row blocks
    const bool pre = p.w1x != nullptr;
    const float* vtab = lds + 2 * SM_BUF;
    const float* vt0 = vtab + (size_t)((wm * 64 + lr) / p.ns) * p.C + lk * 8;
    const float* vt1 = vtab + (size_t)((wm * 64 + 32 + lr) / p.ns) * p.C + lk * 8;
    auto run = [&](const float* A, int nkt, int l, int kt0, int ncol0, int ncols, const float* next_bp, bool subv = false) {
        if (ncols <= 0) { preload(next_bp); return; }
        const float* bp = wptr(l, kt0, ncol0);
        const size_t st = (size_t)NP(l) * 16;
        if (subv) {      // A operand = relu(u - v): the hoisted first layer's centre term and activation
            if (ncols > 32) mfma_ktiles<true, true>(A, nkt, bp, st, a_off, acc, bpre, next_bp, vt0 + kt0 * 16, vt1 + kt0 * 16);
            else mfma_ktiles<false, true>(A, nkt, bp, st, a_off, acc, bpre, next_bp, vt0 + kt0 * 16, vt1 + kt0 * 16);
            return;
        }
        if (ncols > 32) mfma_ktiles<true>(A, nkt, bp, st, a_off, acc, bpre, next_bp);
        else mfma_ktiles<false>(A, nkt, bp, st, a_off, acc, bpre, next_bp);
    };
    auto hidden_epilogue = [&](int l, float* Y) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = wn * 64 + j * 32 + lr;
            if (wn * 64 + j * 32 >= KP(l + 1)) continue;     // wave-uniform: columns the next layer never reads
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    // accumulator r = 4 rq + t  <->  row 8 rq + 4 lk + t of the 32-row block: 4 consecutive rows
                    float4 v;
                    v.x = fmaxf(acc[i][j][4 * rq + 0], 0.f); v.y = fmaxf(acc[i][j][4 * rq + 1], 0.f);
                    v.z = fmaxf(acc[i][j][4 * rq + 2], 0.f); v.w = fmaxf(acc[i][j][4 * rq + 3], 0.f);
                    *reinterpret_cast<float4*>(Y + (size_t)col * SM_LDP + wm * 64 + i * 32 + 8 * rq + 4 * lk) = v;
                }
        }
    };

    preload(wptr(0, 0, wn * 64));
    init_acc(bs0, wn * 64);
    lds_barrier();                                                   // B0
    int cur = 0;   // buffer the next consumer reads
    for (int it = 0; it < sch.n_local; ++it) {
        const bool has_next = it + 1 < sch.n_local;
        int bi, row0;
        sch.tile(it, bi, row0);
        int l = 0;
        JM_SA_STAMP(0);
        if (!single) {
            // ---------------- layer 1: chunks of <= 128 input channels (the gather waves stay one chunk ahead)
            for (int c = 0; c < nchunks; ++c) {
                const int kc = min(SM_KC, K0 - c * SM_KC);     // multiple of 16
                const bool more = c + 1 < nchunks;
                run(lds + cur * SM_BUF, kc / 16, 0, c * (SM_KC / 16), wn * 64, kp1 - wn * 64,
                    more ? wptr(0, (c + 1) * (SM_KC / 16), wn * 64) : wptr(1, 0, wn * 64), pre);
                if (more) { lds_barrier(); cur ^= 1; }
            }
            // the other buffer held chunk nchunks-2 (or the previous tile's last input): every wave left it
            // before the last barrier
            JM_SA_STAMP(1);
            hidden_epilogue(0, lds + (cur ^ 1) * SM_BUF);
            init_acc(BS(1), wn * 64);                          // next layer's bias, set while waiting
            JM_SA_STAMP(2);
            lds_barrier();
            JM_SA_STAMP(3);
            cur ^= 1;
            // ---------------- hidden layers 2 .. L-1
            for (l = 1; l < L - 1; ++l) {
                run(lds + cur * SM_BUF, KP(l) / 16, l, 0, wn * 64, KP(l + 1) - wn * 64, wptr(l + 1, 0, wn * 64));
                hidden_epilogue(l, lds + (cur ^ 1) * SM_BUF);
                init_acc(BS(l + 1), wn * 64);
                lds_barrier();
                cur ^= 1;
            }
        }
        // ---------------- last layer (l == L - 1): 128-column tiles, max-pool epilogue
        {
            const float* A = lds + cur * SM_BUF;
            const int nkt = KP(l) / 16;                        // single: host guarantees one chunk
            const float* bl = BS(l);
            const int npl = NP(l), kpl1 = KP(l + 1);
            for (int n0 = 0; n0 < npl; n0 += 128) {
                const bool last_n = n0 + 128 >= npl;
                run(A, nkt, l, 0, n0 + wn * 64, kpl1 - (n0 + wn * 64),
                    last_n ? wptr(0, 0, wn * 64) : wptr(l, 0, n0 + 128 + wn * 64));
                JM_SA_STAMP(4);
                // max over each centre's nsample rows, straight from the accumulator layout
                // row(i, r, lk) = 32 i + (r & 3) + 8 (r >> 2) + 4 lk  within this wave's 64 rows
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int col = n0 + wn * 64 + j * 32 + lr;
                    if (n0 + wn * 64 + j * 32 >= p.cout) continue;   // wave-uniform
                    float v[4];   // up to 4 centres per wave (nsample 16)
                    if (p.ns == 64) {
                        float t = -INFINITY;
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int r = 0; r < 16; ++r) t = fmaxf(t, acc[i][j][r]);
                        v[0] = t; v[1] = v[2] = v[3] = -INFINITY;
                    } else if (p.ns == 32) {
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            float t = -INFINITY;
#pragma unroll
                            for (int r = 0; r < 16; ++r) t = fmaxf(t, acc[i][j][r]);
                            v[i] = t;
                        }
                        v[2] = v[3] = -INFINITY;
                    } else {   // 16: rows 0-15 of a 32-block are r < 8, rows 16-31 are r >= 8
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            float t0 = -INFINITY, t1 = -INFINITY;
#pragma unroll
                            for (int r = 0; r < 8; ++r) { t0 = fmaxf(t0, acc[i][j][r]); t1 = fmaxf(t1, acc[i][j][r + 8]); }
                            v[2 * i] = t0; v[2 * i + 1] = t1;
                        }
                    }
                    const int per_wave = 64 / p.ns;   // centres per wave-row-block
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        if (c < per_wave) {
                            float t = fmaxf(v[c], __shfl_xor(v[c], 32));   // the other lane half holds the rows + 4
                            const int m = (row0 + wm * 64) / p.ns + c;
                            if (lk == 0 && col < p.cout)
                                p.out[(size_t)bi * p.obs + (size_t)col * p.M + m] = fmaxf(t, 0.f);
                        }
                    }
                }
                // next column tile of this layer, or the next tile's first layer
                if (last_n) init_acc(bs0, wn * 64); else init_acc(bl, n0 + 128 + wn * 64);
            }
            JM_SA_STAMP(5);
            if (has_next) { lds_barrier(); cur ^= 1; }         // the gather waves parked the next tile's chunk 0
            JM_SA_STAMP(6);
        }
    }
}


__global__ void __launch_bounds__(512)
sa_mlp_kernel(SaMlpParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    if (__builtin_amdgcn_readfirstlane(tid >> 8)) sa_gather_role(p, lds, tid - 256);
    else sa_mfma_role(p, lds, tid);
}

}  // namespace jm

namespace jm {
int sa_mlp_wide_launch(int b, int n, int m, int c, int nsample, const float* xyz, const float* new_xyz,
                       const float* features, const int* idx, int L, const int* widths, const float* const* weights,
                       const float* const* biases, float* out, size_t obs, hipStream_t s, const int* cls_count = nullptr,
                       const int* glist = nullptr);                                                 // sa_mlp_wide.hip
const char* sa_wide_unsupported(long long b, int n, int m, int c, int nsample, int group_all, int L, const int* widths);
bool sa_xyz_valu_supported(int m, int c, int nsample, int num_layers, const int* widths);                          // sa_xyz.hip
int sa_xyz_valu_launch(int b, int n, int m, int nsample, const float* xyz, const float* new_xyz, const int* idx,
                       const int* widths, const float* const* weights, const float* const* biases, float* out, size_t obs, hipStream_t s,
                       const int* cls_count = nullptr, const int* glist = nullptr);
}

using namespace jm;

extern "C" int jm_sa_mlp_supported(int b, int n, int m, int c, int nsample, int group_all, int num_layers,
                                   const int* widths) {
    if (b < 0 || n < 1 || m < 0 || c < 0 || num_layers < 1 || !widths || widths[0] != 3 + c) return 0;
    // every launch-time requirement of sa_mlp_narrow_launch: "supported" must imply that the launch succeeds
    bool narrow = !group_all && (nsample == 16 || nsample == 32 || nsample == 64) && ((long long)m * nsample) % SM_BM == 0 &&
                  num_layers <= 4 && (num_layers > 1 || sa_first_kp(widths[0]) <= SM_KC) &&
                  (long long)m * nsample / SM_BM * b < (1LL << 31) && widths[num_layers] >= 1;
    for (int l = 1; l < num_layers && narrow; ++l) narrow = widths[l] >= 1 && widths[l] <= 128;
    if (narrow && !group_all && sa_xyz_valu_supported(m, c, nsample, num_layers, widths)) return 3;   // xyz-only scale: sa_xyz.hip
    if (narrow) return 1;
    return sa_wide_unsupported(b, n, m, c, nsample, group_all, num_layers, widths) == nullptr ? 2 : 0;
}

extern "C" size_t jm_sa_mlp_packed_weight_elems(int cout, int cin, int first_layer) {
    if (cout < 1 || cin < 1 || (first_layer && cin < 3)) return 0;
    return (size_t)(first_layer ? sa_first_kp(cin) : pad_to(cin, 16)) * pad_to(cout, 128) + sa_kmajor_elems(cout, cin);
}

extern "C" size_t jm_sa_mlp_packed_bias_elems(int cout) { return cout < 1 ? 0 : (size_t)pad_to(cout, 128); }

extern "C" int jm_sa_mlp_pack(int cout, int cin, int first_layer, const float* w, const float* b, float* wp, float* bp,
                              jm_stream_t stream) {
    JM_REQUIRE(cout >= 1 && cin >= 1 && (!first_layer || cin >= 3), "sa_mlp_pack: bad sizes");
    JM_REQUIRE(w && wp && bp, "sa_mlp_pack: null pointer");
    const int Kp = first_layer ? sa_first_kp(cin) : pad_to(cin, 16), Np = pad_to(cout, 128);
    const long long total = (long long)Kp * Np;
    hipLaunchKernelGGL(sa_mlp_pack_kernel, dim3((unsigned)divup(total, 256)), dim3(256), 0, (hipStream_t)stream, cout, cin,
                       Kp, Np, first_layer ? 1 : 0, w, b, wp, bp);
    return check_launch("sa_mlp_pack");
}

/* layer widths: widths[0] = 3 + C (input), widths[1..L] = layer outputs.
 * weights[l] / biases[l]: packed by jm_sa_mlp_pack(widths[l+1], widths[l], l == 0, ...). */
static int sa_mlp_narrow_launch(int b, int n, int m, int c, int nsample, const float* xyz, const float* new_xyz,
                                const float* features, const float* w1x, const int* idx, int num_layers, const int* widths,
                                const float* const* weights, const float* const* biases, float* out, size_t obs, jm_stream_t stream);
#ifdef JM_TOOLS_BUILD
static long long* g_sa_trace = nullptr;      // tools build only: the product library keeps no state
extern "C" __attribute__((visibility("default"))) void jm_tools_set_sa_trace(long long* buf) { g_sa_trace = buf; }
#endif

/* pre-projected form (see the header): layers 2..L on relu(u[idx] - v[centre]) */
static int check_out_stride(size_t obs, int m, int cout) {
    JM_REQUIRE(obs == 0 || obs >= (size_t)cout * (size_t)m, "sa_mlp: output frame stride below cout * npoint");
    return JM_OK;
}

extern "C" int jm_sa_mlp_forward_pre_into(int b, int n, int m, int c, int nsample, const float* u, const float* w1x,
                                          const float* new_xyz, const int* idx, int num_layers, const int* widths,
                                          const float* const* weights, const float* const* biases, float* out,
                                          size_t out_frame_stride, jm_stream_t stream) {
    JM_REQUIRE(b >= 0 && n >= 1 && m >= 0 && c >= 16 && c % 16 == 0 && c <= 128, "sa_mlp_pre: C must be a multiple of 16, <= 128");
    if (b == 0 || m == 0) return JM_OK;
    JM_REQUIRE(u && w1x && new_xyz && idx && out && widths && weights && biases, "sa_mlp_pre: null pointer");
    JM_REQUIRE(num_layers >= 2 && num_layers <= 3 && widths[0] == c, "sa_mlp_pre: 2 or 3 layers after the hoisted one, widths[0] == C");
    if (int rc = check_out_stride(out_frame_stride, m, widths[num_layers])) return rc;
    return sa_mlp_narrow_launch(b, n, m, c, nsample, nullptr, new_xyz, u, w1x, idx, num_layers, widths, weights, biases, out,
                                out_frame_stride, stream);
}

extern "C" int jm_sa_mlp_forward_pre(int b, int n, int m, int c, int nsample, const float* u, const float* w1x,
                                     const float* new_xyz, const int* idx, int num_layers, const int* widths,
                                     const float* const* weights, const float* const* biases, float* out, jm_stream_t stream) {
    return jm_sa_mlp_forward_pre_into(b, n, m, c, nsample, u, w1x, new_xyz, idx, num_layers, widths, weights, biases, out, 0, stream);
}

extern "C" int jm_sa_mlp_forward(int b, int n, int m, int c, int nsample, const float* xyz, const float* new_xyz,
                                 const float* features, const int* idx, int num_layers, const int* widths,
                                 const float* const* weights, const float* const* biases, float* out,
                                 jm_stream_t stream) {
    return jm_sa_mlp_forward_into(b, n, m, c, nsample, xyz, new_xyz, features, idx, num_layers, widths, weights, biases, out, 0, stream);
}

extern "C" int jm_sa_mlp_forward_into(int b, int n, int m, int c, int nsample, const float* xyz, const float* new_xyz,
                                      const float* features, const int* idx, int num_layers, const int* widths,
                                      const float* const* weights, const float* const* biases, float* out,
                                      size_t out_frame_stride, jm_stream_t stream) {
    JM_REQUIRE(b >= 0 && n >= 1 && m >= 0 && c >= 0, "sa_mlp: bad sizes");
    if (b == 0 || m == 0) return JM_OK;
    JM_REQUIRE(xyz && out && widths && weights && biases && (features || c == 0), "sa_mlp: null pointer");
    JM_REQUIRE((idx == nullptr) == (new_xyz == nullptr), "sa_mlp: idx and new_xyz are both given or both NULL (GroupAll)");
    JM_REQUIRE(num_layers >= 1 && widths[0] == 3 + c, "sa_mlp: widths[0] = %d != 3 + C = %d", widths[0], 3 + c);
    if (int rc = check_out_stride(out_frame_stride, idx ? m : 1, widths[num_layers])) return rc;
    if (jm_sa_mlp_supported(b, n, m, c, nsample, idx == nullptr, num_layers, widths) == 2)   // wide / GroupAll variant
        return sa_mlp_wide_launch(b, n, m, c, nsample, xyz, new_xyz, features, idx, num_layers, widths, weights, biases,
                                  out, out_frame_stride, (hipStream_t)stream);
    JM_REQUIRE(idx && new_xyz, "sa_mlp: GroupAll (idx == NULL) needs a shape of the wide variant");
    JM_REQUIRE(widths[0] == 3 + c, "sa_mlp: widths[0] = %d != 3 + C = %d", widths[0], 3 + c);
    if (sa_xyz_valu_supported(m, c, nsample, num_layers, widths))        // xyz-only scales: the vector pipe (sa_xyz.hip)
        return sa_xyz_valu_launch(b, n, m, nsample, xyz, new_xyz, idx, widths, weights, biases, out, out_frame_stride, (hipStream_t)stream);
    return sa_mlp_narrow_launch(b, n, m, c, nsample, xyz, new_xyz, features, nullptr, idx, num_layers, widths, weights, biases,
                                out, out_frame_stride, stream);
}

static int sa_mlp_narrow_launch(int b, int n, int m, int c, int nsample, const float* xyz, const float* new_xyz,
                                const float* features, const float* w1x, const int* idx, int num_layers, const int* widths,
                                const float* const* weights, const float* const* biases, float* out, size_t obs, jm_stream_t stream) {
    const bool pre = w1x != nullptr;
    JM_REQUIRE(nsample == 16 || nsample == 32 || nsample == 64, "sa_mlp: nsample %d not in {16,32,64}", nsample);
    JM_REQUIRE(((long long)m * nsample) % SM_BM == 0, "sa_mlp: npoint*nsample = %lld is not a multiple of 128", (long long)m * nsample);
    JM_REQUIRE(num_layers >= 1 && num_layers <= 4, "sa_mlp: %d layers unsupported", num_layers);
    // (persistent kernel, grid = #CUs: the batch is bounded only by the 31-bit tile counter below)
    JM_REQUIRE(num_layers > 1 || (pre ? c : sa_first_kp(widths[0])) <= SM_KC, "sa_mlp: a single layer needs C <= 128");
    SaMlpParams p{};
    p.N = n; p.M = m; p.C = c; p.ns = nsample;
    p.xyz = xyz; p.new_xyz = new_xyz; p.feat = features; p.idx = idx; p.w1x = w1x;
    p.L = num_layers;
    for (int l = 0; l <= num_layers; ++l) {
        JM_REQUIRE(widths[l] >= 1, "sa_mlp: bad width");
        JM_REQUIRE(l == 0 || l == num_layers || widths[l] <= 128, "sa_mlp: hidden width %d > 128", widths[l]);
        p.kp[l] = l == 0 ? (pre ? pad_to(c, 16) : sa_first_kp(widths[0])) : pad_to(widths[l], 16);
    }
    for (int l = 0; l < num_layers; ++l) {
        JM_REQUIRE(weights[l] && biases[l], "sa_mlp: null layer %d", l);
        JM_REQUIRE((reinterpret_cast<uintptr_t>(weights[l]) & 15u) == 0, "sa_mlp: weights must be 16-byte aligned");
        p.W[l] = weights[l]; p.bias[l] = biases[l];
        p.np[l] = pad_to(widths[l + 1], 128);
    }
    p.out = out; p.cout = widths[num_layers];
    p.obs = obs ? obs : (size_t)p.cout * (size_t)m;
    const size_t lds_bytes = pre ? SM_LDS_BYTES_PRE : SM_LDS_BYTES;
    (void)hipFuncSetAttribute((const void*)sa_mlp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SM_LDS_BYTES_PRE);
    // persistent: one workgroup per CU (135 KB of LDS each), whole frames per XCD when there are enough of them
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8) cus = 256;
    cus -= cus % 8;
    p.tiles_per_frame = (int)((long long)m * nsample / SM_BM);
    const long long total = (long long)p.tiles_per_frame * b;
    JM_REQUIRE(total < (1LL << 31), "sa_mlp: too many tiles");
    p.total_tiles = (int)total;
    p.xcd_frames = b >= 16 ? 1 : 0;
#ifdef JM_TOOLS_BUILD
    p.trace = g_sa_trace;
    p.dbg = tune_env("JM_SA_DBG", 0);
#endif
    const int grid = p.xcd_frames ? cus : (int)(total < cus ? total : cus);
    hipLaunchKernelGGL(sa_mlp_kernel, dim3((unsigned)grid), dim3(512), lds_bytes, (hipStream_t)stream, p);
    return check_launch("sa_mlp");
}
