// rows_chain.hip — whole CHAINS of the training path's row kernels behind one C call each: a dense-layer stack (forward / backward)
// and one set-abstraction scale (forward / backward).
//
// Why: the joint-mode step (tools/train.py:96-107 without cfg.TRAIN.FINETUNE) is ~600 launches of rows_gemm.hip / rows_ops.hip
// kernels, and issued one by one from Python each costs the host ~17 us (argument marshalling, tensor allocation, autograd
// bookkeeping) against ~4 us for the launch itself: the 4-frame step was host-bound at 25 ms of enqueue.  Here the layer loop of
// pytorch_utils.py:6-33 (SharedMLP) / pointnet2_modules.py:46-55 (one scale of a set-abstraction level) and of their backward runs
// on the host in C++, the caller hands in every buffer (outputs, saved activations, two scratch row buffers, the wgrad
// workspace) and nothing is allocated or synchronised.  The kernels are exactly the single-layer entries' (jm_rows_linear_*,
// jm_sa_rows_*): same bits.
#include "jm_rows.h"
#ifdef JM_TOOLS_BUILD
#include "jmodt_hip_tools.h"
#endif

namespace jm {

// out = dy * (1 - y^2): the gradient through tanh given its OUTPUT y (out may be dy)
__global__ void __launch_bounds__(256)
rows_tanh_grad_kernel(int M, const int* __restrict__ m_dev, int N, const float* dy, int ldd, const float* __restrict__ y, int ldy, float* out, int ldo) {
    const int Mv = dev_count(M, m_dev);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < (long long)Mv * N; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / N), c = (int)(i % N);
        const float v = y[(size_t)r * ldy + c];
        out[(size_t)r * ldo + c] = dy[(size_t)r * ldd + c] * (1.f - v * v);
    }
}

// dst[r, c0 .. c0 + n) = src[r, 0 .. n)
__global__ void rows_copy_cols_kernel(int M, int n, const float* __restrict__ src, int lds, float* __restrict__ dst, int ldd, int c0) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * n) return;
    const int r = i / n, c = i % n;
    dst[(size_t)r * ldd + c0 + c] = src[(size_t)r * lds + c];
}

}  // namespace jm

using namespace jm;

#define JM_TRY(call)            \
    do {                        \
        const int rc_ = (call); \
        if (rc_ != JM_OK) return rc_; \
    } while (0)

extern "C" {

int jm_rows_tanh_grad(int m, const int* m_dev, int n, const float* dy, int ldd, const float* y, int ldy, float* out, int ldo, jm_stream_t stream) {
    JM_REQUIRE(m >= 0 && n > 0 && dy && y && out, "rows_tanh_grad: bad arguments");
    if (m == 0) return JM_OK;
    hipLaunchKernelGGL(rows_tanh_grad_kernel, dim3((unsigned)grid_for((long long)m * n, 256, JM_EGRID)), dim3(256), 0, (hipStream_t)stream, m, m_dev, n, dy,
                       ldd, y, ldy, out, ldo);
    return check_launch("rows_tanh_grad");
}

int jm_rows_mlp_forward(const jm_rows_mlp_t* d, jm_stream_t stream) {
    JM_REQUIRE(d && d->nl >= 1 && d->nl <= JM_ROWS_MAX_LAYERS, "rows_mlp_forward: 1 .. %d layers", JM_ROWS_MAX_LAYERS);
    const float *cur = d->x1, *cur2 = d->k2 ? d->x2 : nullptr;
    int k1 = d->k1, k2 = d->k2, ld1 = d->ldx1, ld2 = d->ldx2;
    for (int l = 0; l < d->nl; ++l) {
        JM_TRY(jm_rows_linear_forward(d->m, d->m_dev, k1, k2, d->widths[l], cur, ld1, cur2, ld2, d->w[l], d->ldw[l], d->b[l], d->acts[l], nullptr,
                                      d->y[l], d->widths[l], stream));
        cur = d->y[l]; cur2 = nullptr; k1 = ld1 = d->widths[l]; k2 = ld2 = 0;
    }
    return JM_OK;
}

int jm_rows_mlp_backward(const jm_rows_mlp_t* d, const jm_rows_mlp_grad_t* g, jm_stream_t stream) {
    JM_REQUIRE(d && g && d->nl >= 1 && d->nl <= JM_ROWS_MAX_LAYERS && g->dout && g->scratch[0] && g->scratch[1], "rows_mlp_backward: bad arguments");
    const int last = d->nl - 1, m = d->m;
    const float* dy = g->dout;
    int lddy = g->lddout, tog = 0;
    // gradient w.r.t. the last layer's pre-activation
    if (d->acts[last] == 1) {
        JM_TRY(jm_rows_relu_mask(m, d->m_dev, d->widths[last], dy, lddy, d->y[last], d->widths[last], g->scratch[0], d->widths[last], stream));
        dy = g->scratch[0]; lddy = d->widths[last]; tog = 1;
    } else if (d->acts[last] == 2) {
        JM_TRY(jm_rows_tanh_grad(m, d->m_dev, d->widths[last], dy, lddy, d->y[last], d->widths[last], g->scratch[0], d->widths[last], stream));
        dy = g->scratch[0]; lddy = d->widths[last]; tog = 1;
    }
    for (int l = last; l >= 1; --l) {
        const int n = d->widths[l], k = d->widths[l - 1];
        const float* x = d->y[l - 1];
        JM_TRY(jm_rows_linear_wgrad(m, d->m_dev, n, k, dy, lddy, x, k, g->dw[l], g->lddw[l], g->db[l], 0, g->ws, g->ws_bytes, stream));
        float* nxt = g->scratch[tog];
        JM_TRY(jm_rows_linear_dgrad(m, d->m_dev, n, k, dy, lddy, d->w[l], d->ldw[l], d->acts[l - 1] == 1 ? x : nullptr, k, 0, nxt, k, stream));
        if (d->acts[l - 1] == 2) JM_TRY(jm_rows_tanh_grad(m, d->m_dev, k, nxt, k, x, k, nxt, k, stream));
        dy = nxt; lddy = k; tog ^= 1;
    }
    const int n = d->widths[0];
    JM_TRY(jm_rows_linear_wgrad(m, d->m_dev, n, d->k1, dy, lddy, d->x1, d->ldx1, g->dw[0], g->lddw[0], g->db[0], 0, g->ws, g->ws_bytes, stream));
    if (d->k2)
        JM_TRY(jm_rows_linear_wgrad(m, d->m_dev, n, d->k2, dy, lddy, d->x2, d->ldx2, g->dw[0] + d->k1, g->lddw[0], nullptr, 0, g->ws, g->ws_bytes, stream));
    if (g->dx1) JM_TRY(jm_rows_linear_dgrad(m, d->m_dev, n, d->k1, dy, lddy, d->w[0], d->ldw[0], nullptr, 0, 0, g->dx1, d->k1, stream));
    if (g->dx2 && d->k2)
        JM_TRY(jm_rows_linear_dgrad(m, d->m_dev, n, d->k2, dy, lddy, d->w[0] + d->k1, d->ldw[0], nullptr, 0, 0, g->dx2, d->k2, stream));
    return JM_OK;
}

#ifdef JM_TOOLS_BUILD   // (tools/csrc/jmodt_hip_tools.h: measured slower on side streams than inline, not in the product ABI)
// ---- the backward in TWO phases (round 5: weight gradients off the critical path).  Nothing downstream waits for a weight gradient
// until the optimizer, but on one stream every layer's wgrad (+ its split-K reduce) sits between two links of the data-gradient chain
// the NEXT module's backward waits for: 97 wgrad launches of ~60 us per joint-mode step.  Phase 1 (`_chain`) runs the data-gradient chain
// only and keeps every layer's pre-activation gradient dys[l] (m, widths[l]); phase 2 (`_wgrads`) computes the weight / bias
// gradients from them and from the saved activations — on whatever stream the caller orders behind phase 1.  Same kernels, same bits.
static const float* mlp_dy(const jm_rows_mlp_t* d, const jm_rows_mlp_grad_t* g, float* const* dys, int l, int* ld) {
    if (l == d->nl - 1 && d->acts[l] == 0) { *ld = g->lddout; return g->dout; }      // no activation behind the last layer: dout IS it
    *ld = d->widths[l];
    return dys[l];
}

int jm_rows_mlp_backward_chain(const jm_rows_mlp_t* d, const jm_rows_mlp_grad_t* g, float* const* dys, jm_stream_t stream) {
    JM_REQUIRE(d && g && dys && d->nl >= 1 && d->nl <= JM_ROWS_MAX_LAYERS && g->dout, "rows_mlp_backward_chain: bad arguments");
    const int last = d->nl - 1, m = d->m;
    if (d->acts[last] == 1)
        JM_TRY(jm_rows_relu_mask(m, d->m_dev, d->widths[last], g->dout, g->lddout, d->y[last], d->widths[last], dys[last], d->widths[last], stream));
    else if (d->acts[last] == 2)
        JM_TRY(jm_rows_tanh_grad(m, d->m_dev, d->widths[last], g->dout, g->lddout, d->y[last], d->widths[last], dys[last], d->widths[last], stream));
    for (int l = last; l >= 1; --l) {
        const int n = d->widths[l], k = d->widths[l - 1];
        int lddy;
        const float* dy = mlp_dy(d, g, dys, l, &lddy);
        const float* x = d->y[l - 1];
        JM_TRY(jm_rows_linear_dgrad(m, d->m_dev, n, k, dy, lddy, d->w[l], d->ldw[l], d->acts[l - 1] == 1 ? x : nullptr, k, 0, dys[l - 1], k, stream));
        if (d->acts[l - 1] == 2) JM_TRY(jm_rows_tanh_grad(m, d->m_dev, k, dys[l - 1], k, x, k, dys[l - 1], k, stream));
    }
    int lddy;
    const float* dy = mlp_dy(d, g, dys, 0, &lddy);
    const int n = d->widths[0];
    if (g->dx1) JM_TRY(jm_rows_linear_dgrad(m, d->m_dev, n, d->k1, dy, lddy, d->w[0], d->ldw[0], nullptr, 0, 0, g->dx1, d->k1, stream));
    if (g->dx2 && d->k2)
        JM_TRY(jm_rows_linear_dgrad(m, d->m_dev, n, d->k2, dy, lddy, d->w[0] + d->k1, d->ldw[0], nullptr, 0, 0, g->dx2, d->k2, stream));
    return JM_OK;
}

int jm_rows_mlp_backward_wgrads(const jm_rows_mlp_t* d, const jm_rows_mlp_grad_t* g, float* const* dys, jm_stream_t stream) {
    JM_REQUIRE(d && g && dys && d->nl >= 1 && d->nl <= JM_ROWS_MAX_LAYERS && g->dout, "rows_mlp_backward_wgrads: bad arguments");
    const int m = d->m;
    for (int l = d->nl - 1; l >= 1; --l) {
        int lddy;
        const float* dy = mlp_dy(d, g, dys, l, &lddy);
        const int n = d->widths[l], k = d->widths[l - 1];
        JM_TRY(jm_rows_linear_wgrad(m, d->m_dev, n, k, dy, lddy, d->y[l - 1], k, g->dw[l], g->lddw[l], g->db[l], 0, g->ws, g->ws_bytes, stream));
    }
    int lddy;
    const float* dy = mlp_dy(d, g, dys, 0, &lddy);
    const int n = d->widths[0];
    JM_TRY(jm_rows_linear_wgrad(m, d->m_dev, n, d->k1, dy, lddy, d->x1, d->ldx1, g->dw[0], g->lddw[0], g->db[0], 0, g->ws, g->ws_bytes, stream));
    if (d->k2)
        JM_TRY(jm_rows_linear_wgrad(m, d->m_dev, n, d->k2, dy, lddy, d->x2, d->ldx2, g->dw[0] + d->k1, g->lddw[0], nullptr, 0, g->ws, g->ws_bytes, stream));
    return JM_OK;
}

int jm_sa_scale_backward_chain(const jm_sa_scale_t* d, const jm_sa_scale_grad_t* g, float* const* dys, jm_stream_t stream) {
    JM_REQUIRE(d && g && dys && d->nl >= 2 && d->nl <= JM_ROWS_MAX_LAYERS && g->dout, "sa_scale_backward_chain: bad arguments");
    const int R = d->max_rows, C = d->widths[d->nl - 1], H1 = d->widths[0];
    JM_TRY(jm_sa_rows_pool_grad(R, d->rows_dev, C, g->dout, g->lddout, d->out, d->ldo, d->argrow, d->row_group, dys[d->nl - 1], C, stream));
    for (int l = d->nl - 1; l >= 1; --l) {
        const int n = d->widths[l], k = d->widths[l - 1];
        JM_TRY(jm_rows_linear_dgrad(R, d->rows_dev, n, k, dys[l], n, d->w[l], k, d->h[l - 1], k, 0, dys[l - 1], k, stream));
    }
    if (d->c > 0) {
        JM_REQUIRE(g->du, "sa_scale_backward_chain: du");
        (void)jm_zero_async(g->du, (size_t)d->points * H1 * sizeof(float), (hipStream_t)stream);
        JM_TRY(jm_sa_rows_scatter_add(R, d->rows_dev, H1, dys[0], H1, d->row_point, g->du, H1, stream));
        if (g->df) JM_TRY(jm_rows_linear_dgrad(d->points, nullptr, H1, d->c, g->du, H1, d->w1f, d->c, nullptr, 0, g->df_accumulate, g->df, d->c, stream));
    }
    return check_launch("sa_scale_backward_chain");
}

int jm_sa_scale_backward_wgrads(const jm_sa_scale_t* d, const jm_sa_scale_grad_t* g, float* const* dys, jm_stream_t stream) {
    JM_REQUIRE(d && g && dys && d->nl >= 2 && d->nl <= JM_ROWS_MAX_LAYERS && g->dw1 && g->db1 && g->dw4, "sa_scale_backward_wgrads: bad arguments");
    const int R = d->max_rows, H1 = d->widths[0];
    hipStream_t s = (hipStream_t)stream;
    for (int l = d->nl - 1; l >= 1; --l) {
        const int n = d->widths[l], k = d->widths[l - 1];
        JM_TRY(jm_rows_linear_wgrad(R, d->rows_dev, n, k, dys[l], n, d->h[l - 1], k, g->dw[l], k, g->db[l], 0, g->ws, g->ws_bytes, stream));
    }
    JM_TRY(jm_rows_linear_wgrad(R, d->rows_dev, H1, 4, dys[0], H1, d->delta, 4, g->dw4, 4, g->db1, 0, g->ws, g->ws_bytes, stream));
    const int ldw1 = 3 + d->c;
    hipLaunchKernelGGL(rows_copy_cols_kernel, dim3((unsigned)divup(H1 * 3, 256)), dim3(256), 0, s, H1, 3, (const float*)g->dw4, 4, g->dw1, ldw1, 0);
    if (d->c > 0)
        JM_TRY(jm_rows_linear_wgrad(d->points, nullptr, H1, d->c, g->du, H1, d->f, d->ldf, g->dw1 + 3, ldw1, nullptr, 0, g->ws, g->ws_bytes, stream));
    return check_launch("sa_scale_backward_wgrads");
}

#endif  // JM_TOOLS_BUILD

int jm_sa_scale_forward(const jm_sa_scale_t* d, jm_stream_t stream) {
    JM_REQUIRE(d && d->nl >= 2 && d->nl <= JM_ROWS_MAX_LAYERS && d->groups > 0 && d->xyz && d->w1x && d->b1, "sa_scale_forward: bad arguments");
    const int H1 = d->widths[0], R = d->max_rows;
    if (d->c > 0)       // the feature part of the first layer per POINT: u = W1f f + b1
        JM_TRY(jm_rows_linear_forward(d->points, nullptr, d->c, 0, H1, d->f, d->ldf, nullptr, 0, d->w1f, d->c, d->b1, 0, nullptr, d->u, H1, stream));
    JM_TRY(jm_sa_rows_h1(R, d->rows_dev, H1, d->c > 0 ? d->u : nullptr, H1, d->b1, d->w1x, d->xyz, d->ctr, d->row_point, d->row_group, d->h[0], H1,
                         d->delta, stream));
    for (int l = 1; l < d->nl; ++l)
        JM_TRY(jm_rows_linear_forward(R, d->rows_dev, d->widths[l - 1], 0, d->widths[l], d->h[l - 1], d->widths[l - 1], nullptr, 0, d->w[l],
                                      d->widths[l - 1], d->b[l], 1, nullptr, d->h[l], d->widths[l], stream));
    const int C = d->widths[d->nl - 1];
    return jm_sa_rows_pool(d->groups, C, d->h[d->nl - 1], C, d->offsets, d->out, d->ldo, d->argrow, stream);
}

int jm_sa_scale_backward(const jm_sa_scale_t* d, const jm_sa_scale_grad_t* g, jm_stream_t stream) {
    JM_REQUIRE(d && g && d->nl >= 2 && d->nl <= JM_ROWS_MAX_LAYERS && g->dout && g->scratch[0] && g->scratch[1] && g->dw1 && g->db1 && g->dw4,
               "sa_scale_backward: bad arguments");
    const int R = d->max_rows, C = d->widths[d->nl - 1], H1 = d->widths[0];
    hipStream_t s = (hipStream_t)stream;
    JM_TRY(jm_sa_rows_pool_grad(R, d->rows_dev, C, g->dout, g->lddout, d->out, d->ldo, d->argrow, d->row_group, g->scratch[0], C, stream));
    const float* dy = g->scratch[0];
    int tog = 1;
    for (int l = d->nl - 1; l >= 1; --l) {
        const int n = d->widths[l], k = d->widths[l - 1];
        JM_TRY(jm_rows_linear_wgrad(R, d->rows_dev, n, k, dy, n, d->h[l - 1], k, g->dw[l], k, g->db[l], 0, g->ws, g->ws_bytes, stream));
        float* nxt = g->scratch[tog];
        JM_TRY(jm_rows_linear_dgrad(R, d->rows_dev, n, k, dy, n, d->w[l], k, d->h[l - 1], k, 0, nxt, k, stream));
        dy = nxt; tog ^= 1;
    }
    // dy = gradient w.r.t. the first layer's pre-activation on the rows: d(W1x) = dy^T [xyz_j - c_i], d(b1) = its column sums
    JM_TRY(jm_rows_linear_wgrad(R, d->rows_dev, H1, 4, dy, H1, d->delta, 4, g->dw4, 4, g->db1, 0, g->ws, g->ws_bytes, stream));
    const int ldw1 = 3 + d->c;
    hipLaunchKernelGGL(rows_copy_cols_kernel, dim3((unsigned)divup(H1 * 3, 256)), dim3(256), 0, s, H1, 3, (const float*)g->dw4, 4, g->dw1, ldw1, 0);
    if (d->c > 0) {
        (void)jm_zero_async(g->du, (size_t)d->points * H1 * sizeof(float), s);
        JM_TRY(jm_sa_rows_scatter_add(R, d->rows_dev, H1, dy, H1, d->row_point, g->du, H1, stream));
        JM_TRY(jm_rows_linear_wgrad(d->points, nullptr, H1, d->c, g->du, H1, d->f, d->ldf, g->dw1 + 3, ldw1, nullptr, 0, g->ws, g->ws_bytes, stream));
        if (g->df) JM_TRY(jm_rows_linear_dgrad(d->points, nullptr, H1, d->c, g->du, H1, d->w1f, d->c, nullptr, 0, g->df_accumulate, g->df, d->c, stream));
    }
    return check_launch("sa_scale_backward");
}

}  // extern "C"
