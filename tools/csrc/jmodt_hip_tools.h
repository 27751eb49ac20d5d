/*
 * jmodt_hip_tools.h — entry points that exist ONLY in tools/bin/libjmodt_hip_tools.so (python -m jmodt_amd.csrc.build --tools, which
 * compiles the product sources with -DJM_TOOLS_BUILD plus tools/csrc/*.hip).  Each was built, tested exact and MEASURED SLOWER than
 * the route the product library takes; they were moved out of include/jmodt_hip.h in round 6 so that the product ABI holds only
 * entries with a default caller or a reference m.def counterpart ("quarantined").
 */
#ifndef JMODT_HIP_TOOLS_H
#define JMODT_HIP_TOOLS_H
#include "jmodt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)

/* The image branch's kernel == stride transposed convolutions (backbone.py:150-157,187-189: DeConv) as GEMMs whose (rows, columns)
 * matrix is stored PIXEL-SHUFFLED: x (m = B h w, c) = the channels-last input map as rows, wt (k k r, c) with
 * wt[(dy k + dx) r + rr][ci] = W[ci][rr][dy][dx], y = the channels-last (B, h k, w k, ctot) map, this level's r channels at coff:
 * y[b][y k + dy][x k + dx][coff + rr] = sum_ci x[(b, y, x)][ci] wt[...][ci] (no bias).  _dgrad: dx (m, c) from dy in that layout;
 * _wgrad: dwt (k k r, c), workspace jm_rows_wgrad_workspace_bytes(m, k k r, c).  c, r, ctot, coff multiples of 4. */
int jm_rows_deconv_forward(int m, int c, int k, int r, int h, int w, const float* x, int ldx, const float* wt, float* y, int ctot, int coff,
                           jm_stream_t stream);
int jm_rows_deconv_dgrad(int m, int c, int k, int r, int h, int w, const float* dy, int ctot, int coff, const float* wt, float* dx, int lddx,
                         jm_stream_t stream);
int jm_rows_deconv_wgrad(int m, int c, int k, int r, int h, int w, const float* dy, int ctot, int coff, const float* x, int ldx, float* dwt,
                         void* ws, size_t ws_bytes, jm_stream_t stream);

/* The backward in two phases (round 5: weight gradients off the critical path): `_chain` = the data-gradient chain only, keeping every
 * layer's pre-activation gradient in dys[l] (m, widths[l]) — a HOST array of nl device pointers, caller-allocated — and writing dx1 /
 * dx2; `_wgrads` = dw / db of every layer from dys and the saved activations (scratch[] unused by both).  The caller orders phase 2
 * behind phase 1 on whatever stream it likes.  dys[nl - 1] is not written when the last layer has no activation (dout is read instead). */
int jm_rows_mlp_backward_chain(const jm_rows_mlp_t* d, const jm_rows_mlp_grad_t* g, float* const* dys, jm_stream_t stream);
int jm_rows_mlp_backward_wgrads(const jm_rows_mlp_t* d, const jm_rows_mlp_grad_t* g, float* const* dys, jm_stream_t stream);

/* jm_sa_scale_backward in two phases, as above: dys[l] (max_rows, widths[l]); `_chain` also produces du and df (the gradient that goes
 * upstream), `_wgrads` every weight / bias gradient (it reads du) */
int jm_sa_scale_backward_chain(const jm_sa_scale_t* d, const jm_sa_scale_grad_t* g, float* const* dys, jm_stream_t stream);
int jm_sa_scale_backward_wgrads(const jm_sa_scale_t* d, const jm_sa_scale_grad_t* g, float* const* dys, jm_stream_t stream);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif
