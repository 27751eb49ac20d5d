// rows_ops.hip — the gather / pool / scatter kernels around csrc/rows_gemm.hip: together they are the TRAINING path (forward
// AND backward) of the set-abstraction, feature-propagation and LI-Fusion attention blocks on row-major activations.
//
// Set abstraction (pointnet2_modules.py:46-61: QueryAndGroup -> SharedMLP -> max_pool2d over the nsample axis), backward
// included, WITHOUT the (B, C, npoint, nsample) tensors of the reference:
//   * a ball-query list ends in copies of its first hit (ball_query_gpu.cu:36-40) and RoI point sets are cyclically padded
//     (roipool3d_kernel.cu:123-160): rows that are exact copies produce exact copies of every activation, max-pool is
//     idempotent, and the sum of the copies' gradients lands on the same weights — so only the DISTINCT (centre, neighbour)
//     pairs of a group become rows (sa_rows_count / scan / fill; "distinct" = by point index after the optional canonical map,
//     decided by comparing every slot with the slots before it: exact for ANY list, not only ball-query-form lists);
//   * layer 1 on the rows is u[p] + W1x (xyz[p] - c[g]) with u = W1f f + b1 computed per POINT by the GEMM (the reference's
//     per-row K = 3 + C product: pointnet2_utils.py:259-269 concatenates [xyz_j - c_i ; f_j] per row);
//   * pool: per group and channel the maximum over its rows and WHICH row (first maximum: max_pool2d's tie rule); the
//     backward routes d(out) to that row, the row-to-point scatter of d(u) is one float atomic per element (the reference's
//     group_points_grad, group_points_gpu.cu:48-86, does the same on nsample x as many elements).
// Feature propagation (pointnet2_modules.py:139-153): three_interpolate and its gradient on rows (interpolate_gpu.cu:77-161 with
// the channel axis innermost: a wave reads / adds 64 consecutive channels of one coarse point).
// LI-Fusion attention (backbone.py:35-81): the gate's sigmoid and the backward through gate, tanh and the gated product.
#include "jm_rows.h"

namespace jm {

// ------------------------------------------------------------------------------------------------ set-abstraction rows
// one wave per group: entry e_j = canonical point of slot j; slot j is kept iff no earlier slot holds the same entry
template <bool FILL>
__global__ void __launch_bounds__(256)
sa_rows_plan_kernel(int G, int ns, const int* __restrict__ idx, const int* __restrict__ canon, int n_per_set, int groups_per_set,
                    int* __restrict__ d, const int* __restrict__ offsets, int* __restrict__ row_point, int* __restrict__ row_group) {
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (g >= G) return;
    const int set = g / groups_per_set;
    int e = -1 - lane;                                   // lanes beyond ns: distinct dummies that nobody matches
    if (lane < ns) {
        e = idx[(size_t)g * ns + lane];
        if (canon) e = canon[(size_t)set * n_per_set + e];
    }
    bool dup = false;
    for (int i = 0; i < ns; ++i) {
        const int ei = __shfl(e, i);
        dup = dup || (i < lane && ei == e);
    }
    const unsigned long long keep = __ballot(lane < ns && !dup);
    if (!FILL) {
        if (lane == 0) d[g] = __popcll(keep);
    } else if (lane < ns && !dup) {
        const int r = offsets[g] + mbcnt(keep);
        row_point[r] = set * n_per_set + e;
        row_group[r] = g;
    }
}

// exclusive scan of d (G) -> offsets (G + 1); one workgroup of 1024 threads, each owning a contiguous chunk
__global__ void __launch_bounds__(1024)
sa_rows_scan_kernel(int G, const int* __restrict__ d, int* __restrict__ offsets) {
    __shared__ int wsum[16];
    __shared__ int total;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (G + 1023) / 1024;
    const int b = tid * per, e = min(G, b + per);
    int s = 0;
    for (int i = b; i < e; ++i) s += d[i];
    const int incl = wave_incl_scan_i32_dpp(s);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int w = 0; w < 16; ++w) { const int v = wsum[w]; wsum[w] = run; run += v; }
        total = run;
    }
    __syncthreads();
    int run = wsum[wave] + incl - s;
    for (int i = b; i < e; ++i) { offsets[i] = run; run += d[i]; }
    if (tid == 0) offsets[G] = total;
}

// h1[r, c] = relu((u ? u[p, c] : b1[c]) + W1x[c, :] . (xyz[p] - ctr[g]))     thread = (row, 4 channels)
__global__ void __launch_bounds__(256)
sa_rows_h1_kernel(int R, const int* __restrict__ r_dev, int H, const float* __restrict__ u, int ldu, const float* __restrict__ b1,
                  const float* __restrict__ w1x, const float* __restrict__ xyz, const float* __restrict__ ctr,
                  const int* __restrict__ row_point, const int* __restrict__ row_group, float* __restrict__ h1, int ldh,
                  float* __restrict__ delta) {
    const int Rv = dev_count(R, r_dev);
    const int q = H / 4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < (long long)Rv * q; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / q), c = (int)(i % q) * 4;
        const int p = row_point[r], g = row_group[r];
        float dx = xyz[(size_t)p * 3 + 0], dy = xyz[(size_t)p * 3 + 1], dz = xyz[(size_t)p * 3 + 2];
        if (ctr) { dx -= ctr[(size_t)g * 3 + 0]; dy -= ctr[(size_t)g * 3 + 1]; dz -= ctr[(size_t)g * 3 + 2]; }
        if (delta != nullptr && c == 0) *reinterpret_cast<float4*>(delta + (size_t)r * 4) = make_float4(dx, dy, dz, 0.f);
        float4 v = u ? *reinterpret_cast<const float4*>(u + (size_t)p * ldu + c) : *reinterpret_cast<const float4*>(b1 + c);
        float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float* w = w1x + (size_t)(c + k) * 3;
            o[k] = fmaxf(o[k] + w[0] * dx + w[1] * dy + w[2] * dz, 0.f);
        }
        *reinterpret_cast<float4*>(h1 + (size_t)r * ldh + c) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// out[g, c] = max over the group's rows of h[r, c], argrow[g, c] = the FIRST row that holds it       thread = (group, channel)
__global__ void __launch_bounds__(256)
sa_rows_pool_kernel(int G, int C, const float* __restrict__ h, int ldh, const int* __restrict__ offsets, float* __restrict__ out, int ldo,
                    int* __restrict__ argrow) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= (long long)G * C) return;
    const int g = (int)(i / C), c = (int)(i % C);
    const int r0 = offsets[g], r1 = offsets[g + 1];
    float best = h[(size_t)r0 * ldh + c];
    int arg = r0;
    for (int r = r0 + 1; r < r1; ++r) {
        const float v = h[(size_t)r * ldh + c];
        if (v > best) { best = v; arg = r; }
    }
    out[(size_t)g * ldo + c] = best;
    argrow[(size_t)g * C + c] = arg;
}

// dh[r, c] = d(out)[g, c] where r is the group's arg-max row of channel c and the pooled value is positive (ReLU), else 0
__global__ void __launch_bounds__(256)
sa_rows_pool_grad_kernel(int R, const int* __restrict__ r_dev, int C, const float* __restrict__ dout, int lddo, const float* __restrict__ out,
                         int ldo, const int* __restrict__ argrow, const int* __restrict__ row_group, float* __restrict__ dh, int ldd) {
    const int Rv = dev_count(R, r_dev);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < (long long)Rv * C; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / C), c = (int)(i % C);
        const int g = row_group[r];
        const bool hit = argrow[(size_t)g * C + c] == r && out[(size_t)g * ldo + c] > 0.f;
        dh[(size_t)r * ldd + c] = hit ? dout[(size_t)g * lddo + c] : 0.f;
    }
}

// du[p, c] += dh1[r, c] (float atomics: rows of different groups meet at a point)
__global__ void __launch_bounds__(256)
sa_rows_scatter_kernel(int R, const int* __restrict__ r_dev, int H, const float* __restrict__ dh1, int ldd, const int* __restrict__ row_point,
                       float* __restrict__ du, int ldu) {
    const int Rv = dev_count(R, r_dev);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < (long long)Rv * H; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / H), c = (int)(i % H);
        const float v = dh1[(size_t)r * ldd + c];
        if (v != 0.f) atomicAdd(du + (size_t)row_point[r] * ldu + c, v);
    }
}

// partial[s, c, a] = sum over the rows of chunk s of dh1[r, c] * (xyz[p_r, a] - ctr[g_r, a]);   workgroup = chunk; thread =
// (channel, row subset): 256 / H subsets stride the chunk's rows and are combined through LDS in subset order
__global__ void __launch_bounds__(256)
sa_rows_xyz_wgrad_kernel(int R, const int* __restrict__ r_dev, int H, int chunks, const float* __restrict__ dh1, int ldd,
                         const float* __restrict__ xyz, const float* __restrict__ ctr, const int* __restrict__ row_point,
                         const int* __restrict__ row_group, float* __restrict__ partial) {
    __shared__ float red[256 * 3];
    const int Rv = dev_count(R, r_dev);
    const int per = (Rv + chunks - 1) / chunks;
    const int r0 = min(Rv, (int)blockIdx.x * per), r1 = min(Rv, r0 + per);
    const int nsub = H >= 256 ? 1 : 256 / H;
    for (int cb = 0; cb < H; cb += 256) {
        const int c = cb + (int)threadIdx.x % min(H, 256), sub = (int)threadIdx.x / min(H, 256);
        float sx = 0.f, sy = 0.f, sz = 0.f;
        if (c < H && sub < nsub)
            for (int r = r0 + sub; r < r1; r += nsub) {
                const int p = row_point[r], g = row_group[r];
                float dx = xyz[(size_t)p * 3 + 0], dy = xyz[(size_t)p * 3 + 1], dz = xyz[(size_t)p * 3 + 2];
                if (ctr) { dx -= ctr[(size_t)g * 3 + 0]; dy -= ctr[(size_t)g * 3 + 1]; dz -= ctr[(size_t)g * 3 + 2]; }
                const float v = dh1[(size_t)r * ldd + c];
                sx += v * dx; sy += v * dy; sz += v * dz;
            }
        red[threadIdx.x * 3 + 0] = sx; red[threadIdx.x * 3 + 1] = sy; red[threadIdx.x * 3 + 2] = sz;
        __syncthreads();
        if (sub == 0 && c < H) {
            const int w = min(H, 256);
            for (int k = 1; k < nsub; ++k) {
                sx += red[((size_t)k * w + threadIdx.x) * 3 + 0]; sy += red[((size_t)k * w + threadIdx.x) * 3 + 1];
                sz += red[((size_t)k * w + threadIdx.x) * 3 + 2];
            }
            float* o = partial + ((size_t)blockIdx.x * H + c) * 3;
            o[0] = sx; o[1] = sy; o[2] = sz;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ feature propagation rows
// out[b n + i, c] = sum_k w[b, i, k] known[b m + idx[b, i, k], c]          thread = (point, 4 channels)
__global__ void __launch_bounds__(256)
three_interpolate_rows_kernel(int B, int n, int m, int C, const float* __restrict__ known, int ldk, const int* __restrict__ idx,
                              const float* __restrict__ w, float* __restrict__ out, int ldo) {
    const int q = C / 4;
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= (long long)B * n * q) return;
    const int pt = (int)(i / q), c = (int)(i % q) * 4;
    const int b = pt / n;
    const int* id = idx + (size_t)pt * 3;
    const float* ww = w + (size_t)pt * 3;
    const float4 a = *reinterpret_cast<const float4*>(known + ((size_t)b * m + id[0]) * ldk + c);
    const float4 d = *reinterpret_cast<const float4*>(known + ((size_t)b * m + id[1]) * ldk + c);
    const float4 e = *reinterpret_cast<const float4*>(known + ((size_t)b * m + id[2]) * ldk + c);
    const float w0 = ww[0], w1 = ww[1], w2 = ww[2];
    // (interpolate_gpu.cu:96: w0 * p0 + w1 * p1 + w2 * p2, left to right)
    *reinterpret_cast<float4*>(out + (size_t)pt * ldo + c) =
        make_float4(w0 * a.x + w1 * d.x + w2 * e.x, w0 * a.y + w1 * d.y + w2 * e.y, w0 * a.z + w1 * d.z + w2 * e.z,
                    w0 * a.w + w1 * d.w + w2 * e.w);
}

// dknown[b m + idx[b, i, k], c] += w[b, i, k] dout[b n + i, c]   (interpolate_gpu.cu:128-161; consecutive lanes = consecutive channels)
__global__ void __launch_bounds__(256)
three_interpolate_rows_grad_kernel(int B, int n, int m, int C, const float* __restrict__ dout, int ldo, const int* __restrict__ idx,
                                   const float* __restrict__ w, float* __restrict__ dknown, int ldk) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= (long long)B * n * C) return;
    const int pt = (int)(i / C), c = (int)(i % C);
    const int b = pt / n;
    const float g = dout[(size_t)pt * ldo + c];
    const int* id = idx + (size_t)pt * 3;
    const float* ww = w + (size_t)pt * 3;
#pragma unroll
    for (int k = 0; k < 3; ++k) atomicAdd(dknown + ((size_t)b * m + id[k]) * ldk + c, g * ww[k]);
}

// ------------------------------------------------------------------------------------------------ element-wise
// out = y > 0 ? dy : 0 (out may be dy)
__global__ void __launch_bounds__(256)
rows_relu_mask_kernel(int M, const int* __restrict__ m_dev, int N, const float* dy, int ldd, const float* __restrict__ y, int ldy, float* out, int ldo) {
    const int Mv = dev_count(M, m_dev);
    const int q = N / 4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < (long long)Mv * q; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / q), c = (int)(i % q) * 4;
        float4 d = *reinterpret_cast<const float4*>(dy + (size_t)r * ldd + c);
        const float4 v = *reinterpret_cast<const float4*>(y + (size_t)r * ldy + c);
        d.x = v.x > 0.f ? d.x : 0.f; d.y = v.y > 0.f ? d.y : 0.f; d.z = v.z > 0.f ? d.z : 0.f; d.w = v.w > 0.f ? d.w : 0.f;
        *reinterpret_cast<float4*>(out + (size_t)r * ldo + c) = d;
    }
}

// g[r] = sigmoid(z[r * ldz])
__global__ void rows_sigmoid_kernel(int M, const float* __restrict__ z, int ldz, float* __restrict__ g) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < M) g[r] = 1.f / (1.f + expf(-z[(size_t)r * ldz]));
}

// backward through  J = relu(Jpre) * g,  g = sigmoid(z),  z = t . w3 + b3,  t = tanh(...)   (backbone.py:54-62), one wave per row:
//   d(gate) = sum_c dJ[c] * Jpre[c] with Jpre = J / g;  dJ <- dJ * g where J > 0 (in place: the gradient w.r.t. conv1's pre-activation)
//   dz[r, 0] = d(gate) g (1 - g), dz[r, 1..3] = 0;  dt[r, k] = dz w3[k] (1 - t[k]^2)
__global__ void __launch_bounds__(256)
rows_gate_backward_kernel(int M, int PC, int RC, float* __restrict__ dj, int ldj, const float* __restrict__ j, int ldjj, const float* __restrict__ g,
                          const float* __restrict__ t, int ldt, const float* __restrict__ w3, float* __restrict__ dz, float* __restrict__ dt, int lddt) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= M) return;
    const float gv = g[r];
    const float inv = gv > 0.f ? 1.f / gv : 0.f;
    float s = 0.f;
    for (int c = lane; c < PC; c += 64) {
        const float d = dj[(size_t)r * ldj + c], v = j[(size_t)r * ldjj + c];
        s += d * (v * inv);
        dj[(size_t)r * ldj + c] = v > 0.f ? d * gv : 0.f;
    }
    s = wave_sum_f32(s);
    const float dzv = s * gv * (1.f - gv);
    if (lane < 4) dz[(size_t)r * 4 + lane] = lane == 0 ? dzv : 0.f;
    for (int k = lane; k < RC; k += 64) {
        const float tv = t[(size_t)r * ldt + k];
        dt[(size_t)r * lddt + k] = dzv * w3[k] * (1.f - tv * tv);
    }
}

// ------------------------------------------------------------------------------------------------ BatchNorm folding, multi-tensor
// Every BatchNorm-followed convolution of the network in ONE launch (the table travels in the kernel arguments, as the fused
// optimizers' multi-tensor kernels do): wf[l] = w[l] * s[soff[l] + row], and the backward dw[l] = dwf[l] * s, ds[row] = sum_col
// dwf[l][row, col] * w[l][row, col].  A layer's (rows, cols) block is contiguous per row in whatever memory format it has.
#define JM_FOLD_MAX 48
struct FoldEntry { const float* a; const float* b; float* o; int rows, cols, soff; };
struct FoldTable { int n; FoldEntry e[JM_FOLD_MAX]; };

__global__ void __launch_bounds__(256)
fold_bn_multi_kernel(FoldTable t, const float* __restrict__ s) {
    const FoldEntry e = t.e[blockIdx.x];
    const long long total = (long long)e.rows * e.cols;
    for (long long i = blockIdx.y * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.y * blockDim.x)
        e.o[i] = e.a[i] * s[e.soff + (int)(i / e.cols)];
}

// one wave per row: a = dwf, b = w, o = dw
__global__ void __launch_bounds__(256)
fold_bn_multi_grad_kernel(FoldTable t, const float* __restrict__ s, float* __restrict__ ds) {
    const FoldEntry e = t.e[blockIdx.x];
    const int lane = threadIdx.x & 63;
    for (int row = blockIdx.y * 4 + (threadIdx.x >> 6); row < e.rows; row += gridDim.y * 4) {
        const float sv = s[e.soff + row];
        const size_t base = (size_t)row * e.cols;
        float acc = 0.f;
        for (int c = lane; c < e.cols; c += 64) {
            const float g = e.a[base + c];
            acc += g * e.b[base + c];
            e.o[base + c] = g * sv;
        }
        acc = wave_sum_f32(acc);
        if (lane == 0) ds[e.soff + row] = acc;
    }
}

}  // namespace jm

using namespace jm;

extern "C" {

int jm_sa_rows_plan(int groups, int ns, const int* idx, const int* canon, int n_per_set, int groups_per_set, int* d, int* offsets,
                    int* row_point, int* row_group, jm_stream_t stream) {
    JM_REQUIRE(groups > 0 && ns > 0 && ns <= 64 && idx && d && offsets && row_point && row_group && n_per_set > 0 && groups_per_set > 0,
               "sa_rows_plan: bad arguments (groups %d, nsample %d <= 64)", groups, ns);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL((sa_rows_plan_kernel<false>), dim3((unsigned)divup(groups, 4)), dim3(256), 0, s, groups, ns, idx, canon, n_per_set,
                       groups_per_set, d, (const int*)nullptr, (int*)nullptr, (int*)nullptr);
    hipLaunchKernelGGL(sa_rows_scan_kernel, dim3(1), dim3(1024), 0, s, groups, (const int*)d, offsets);
    hipLaunchKernelGGL((sa_rows_plan_kernel<true>), dim3((unsigned)divup(groups, 4)), dim3(256), 0, s, groups, ns, idx, canon, n_per_set,
                       groups_per_set, (int*)nullptr, (const int*)offsets, row_point, row_group);
    return check_launch("sa_rows_plan");
}

int jm_sa_rows_h1(int rows, const int* rows_dev, int h, const float* u, int ldu, const float* b1, const float* w1x, const float* xyz,
                  const float* ctr, const int* row_point, const int* row_group, float* h1, int ldh, float* delta, jm_stream_t stream) {
    JM_REQUIRE(rows >= 0 && h > 0 && h % 4 == 0 && (u || b1) && w1x && xyz && row_point && row_group && h1 && ldh % 4 == 0 && (!u || ldu % 4 == 0),
               "sa_rows_h1: bad arguments (h %d, ldu %d, ldh %d)", h, ldu, ldh);
    if (rows == 0) return JM_OK;
    hipLaunchKernelGGL(sa_rows_h1_kernel, dim3((unsigned)grid_for((long long)rows * (h / 4), 256, JM_EGRID)), dim3(256), 0, (hipStream_t)stream, rows,
                       rows_dev, h, u, ldu, b1, w1x, xyz, ctr, row_point, row_group, h1, ldh, delta);
    return check_launch("sa_rows_h1");
}

int jm_sa_rows_pool(int groups, int c, const float* h, int ldh, const int* offsets, float* out, int ldo, int* argrow, jm_stream_t stream) {
    JM_REQUIRE(groups > 0 && c > 0 && h && offsets && out && argrow && ldo >= c, "sa_rows_pool: bad arguments");
    hipLaunchKernelGGL(sa_rows_pool_kernel, dim3((unsigned)grid_for((long long)groups * c)), dim3(256), 0, (hipStream_t)stream, groups, c, h, ldh,
                       offsets, out, ldo, argrow);
    return check_launch("sa_rows_pool");
}

int jm_sa_rows_pool_grad(int rows, const int* rows_dev, int c, const float* dout, int lddo, const float* out, int ldo, const int* argrow,
                         const int* row_group, float* dh, int ldd, jm_stream_t stream) {
    JM_REQUIRE(rows >= 0 && c > 0 && dout && out && argrow && row_group && dh, "sa_rows_pool_grad: bad arguments");
    if (rows == 0) return JM_OK;
    hipLaunchKernelGGL(sa_rows_pool_grad_kernel, dim3((unsigned)grid_for((long long)rows * c, 256, JM_EGRID)), dim3(256), 0, (hipStream_t)stream, rows,
                       rows_dev, c, dout, lddo, out, ldo, argrow, row_group, dh, ldd);
    return check_launch("sa_rows_pool_grad");
}

int jm_sa_rows_scatter_add(int rows, const int* rows_dev, int h, const float* dh1, int ldd, const int* row_point, float* du, int ldu,
                           jm_stream_t stream) {
    JM_REQUIRE(rows >= 0 && h > 0 && dh1 && row_point && du, "sa_rows_scatter_add: bad arguments");
    if (rows == 0) return JM_OK;
    hipLaunchKernelGGL(sa_rows_scatter_kernel, dim3((unsigned)grid_for((long long)rows * h, 256, JM_EGRID)), dim3(256), 0, (hipStream_t)stream, rows,
                       rows_dev, h, dh1, ldd, row_point, du, ldu);
    return check_launch("sa_rows_scatter_add");
}

int jm_sa_rows_xyz_wgrad(int rows, const int* rows_dev, int h, const float* dh1, int ldd, const float* xyz, const float* ctr,
                         const int* row_point, const int* row_group, float* dw1x, int accumulate, void* ws, size_t ws_bytes, jm_stream_t stream) {
    // dw1x (h, 3) (+)= sum_r dh1[r, :]^T (xyz[p_r] - ctr[g_r])
    JM_REQUIRE(rows >= 0 && h > 0 && dh1 && xyz && row_point && row_group && dw1x, "sa_rows_xyz_wgrad: bad arguments");
    if (!ws || ws_bytes < jm_rows_reduce_workspace_bytes(h * 3)) {
        set_error("sa_rows_xyz_wgrad: workspace of %zu bytes, need %zu", ws_bytes, jm_rows_reduce_workspace_bytes(h * 3));
        return JM_EWORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    const int chunks = rows_chunks(rows);
    hipLaunchKernelGGL(sa_rows_xyz_wgrad_kernel, dim3((unsigned)chunks), dim3(256), 0, s, rows, rows_dev, h, chunks, dh1, ldd, xyz, ctr,
                       row_point, row_group, (float*)ws);
    hipLaunchKernelGGL(rows_sum_partials_kernel, dim3((unsigned)divup(h * 3, 256)), dim3(256), 0, s, h * 3, chunks, (const float*)ws, dw1x,
                       accumulate);
    return check_launch("sa_rows_xyz_wgrad");
}

int jm_three_interpolate_rows(int b, int n, int m, int c, const float* known, int ldk, const int* idx, const float* w, float* out, int ldo,
                              jm_stream_t stream) {
    JM_REQUIRE(b > 0 && n > 0 && m > 0 && c > 0 && c % 4 == 0 && ldk % 4 == 0 && ldo % 4 == 0 && known && idx && w && out,
               "three_interpolate_rows: bad arguments (c %d, ldk %d, ldo %d must be multiples of 4)", c, ldk, ldo);
    hipLaunchKernelGGL(three_interpolate_rows_kernel, dim3((unsigned)grid_for((long long)b * n * (c / 4))), dim3(256), 0, (hipStream_t)stream, b, n, m,
                       c, known, ldk, idx, w, out, ldo);
    return check_launch("three_interpolate_rows");
}

int jm_three_interpolate_rows_grad(int b, int n, int m, int c, const float* dout, int ldo, const int* idx, const float* w, float* dknown, int ldk,
                                   jm_stream_t stream) {
    JM_REQUIRE(b > 0 && n > 0 && m > 0 && c > 0 && dout && idx && w && dknown, "three_interpolate_rows_grad: bad arguments");
    hipLaunchKernelGGL(three_interpolate_rows_grad_kernel, dim3((unsigned)grid_for((long long)b * n * c)), dim3(256), 0, (hipStream_t)stream, b, n, m,
                       c, dout, ldo, idx, w, dknown, ldk);
    return check_launch("three_interpolate_rows_grad");
}

int jm_rows_relu_mask(int m, const int* m_dev, int n, const float* dy, int ldd, const float* y, int ldy, float* out, int ldo, jm_stream_t stream) {
    JM_REQUIRE(m >= 0 && n > 0 && n % 4 == 0 && ldd % 4 == 0 && ldy % 4 == 0 && ldo % 4 == 0 && dy && y && out,
               "rows_relu_mask: bad arguments (n %d, ldd %d, ldy %d, ldo %d)", n, ldd, ldy, ldo);
    if (m == 0) return JM_OK;
    hipLaunchKernelGGL(rows_relu_mask_kernel, dim3((unsigned)grid_for((long long)m * (n / 4), 256, JM_EGRID)), dim3(256), 0, (hipStream_t)stream, m, m_dev,
                       n, dy, ldd, y, ldy, out, ldo);
    return check_launch("rows_relu_mask");
}

int jm_rows_sigmoid(int m, const float* z, int ldz, float* g, jm_stream_t stream) {
    JM_REQUIRE(m > 0 && z && g && ldz > 0, "rows_sigmoid: bad arguments");
    hipLaunchKernelGGL(rows_sigmoid_kernel, dim3((unsigned)divup(m, 256)), dim3(256), 0, (hipStream_t)stream, m, z, ldz, g);
    return check_launch("rows_sigmoid");
}

int jm_rows_gate_backward(int m, int pc, int rc, float* dj, int ldj, const float* j, int ldjj, const float* g, const float* t, int ldt,
                          const float* w3, float* dz, float* dt, int lddt, jm_stream_t stream) {
    JM_REQUIRE(m > 0 && pc > 0 && rc > 0 && dj && j && g && t && w3 && dz && dt, "rows_gate_backward: bad arguments");
    hipLaunchKernelGGL(rows_gate_backward_kernel, dim3((unsigned)divup(m, 4)), dim3(256), 0, (hipStream_t)stream, m, pc, rc, dj, ldj, j, ldjj, g, t,
                       ldt, w3, dz, dt, lddt);
    return check_launch("rows_gate_backward");
}

int jm_fold_bn_multi(int n, const float* const* w, float* const* wf, const int* rows, const int* cols, const int* soff, const float* s,
                     jm_stream_t stream) {
    JM_REQUIRE(n >= 0 && (n == 0 || (w && wf && rows && cols && soff && s)), "fold_bn_multi: bad arguments");
    for (int b = 0; b < n; b += JM_FOLD_MAX) {
        FoldTable t{};
        t.n = n - b < JM_FOLD_MAX ? n - b : JM_FOLD_MAX;
        for (int i = 0; i < t.n; ++i) t.e[i] = FoldEntry{w[b + i], nullptr, wf[b + i], rows[b + i], cols[b + i], soff[b + i]};
        hipLaunchKernelGGL(fold_bn_multi_kernel, dim3((unsigned)t.n, 32), dim3(256), 0, (hipStream_t)stream, t, s);
    }
    return check_launch("fold_bn_multi");
}

int jm_fold_bn_multi_grad(int n, const float* const* dwf, const float* const* w, float* const* dw, const int* rows, const int* cols,
                          const int* soff, const float* s, float* ds, jm_stream_t stream) {
    JM_REQUIRE(n >= 0 && (n == 0 || (dwf && w && dw && rows && cols && soff && s && ds)), "fold_bn_multi_grad: bad arguments");
    for (int b = 0; b < n; b += JM_FOLD_MAX) {
        FoldTable t{};
        t.n = n - b < JM_FOLD_MAX ? n - b : JM_FOLD_MAX;
        for (int i = 0; i < t.n; ++i) t.e[i] = FoldEntry{dwf[b + i], w[b + i], dw[b + i], rows[b + i], cols[b + i], soff[b + i]};
        hipLaunchKernelGGL(fold_bn_multi_grad_kernel, dim3((unsigned)t.n, 32), dim3(256), 0, (hipStream_t)stream, t, s, ds);
    }
    return check_launch("fold_bn_multi_grad");
}

}  // extern "C"
