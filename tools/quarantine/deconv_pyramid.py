"""QUARANTINED (round 6): the image branch's kernel == stride transposed convolutions as pixel-shuffled GEMMs
(jm_rows_deconv_* — exported by the TOOLS build only, tools/csrc/jmodt_hip_tools.h).  Exact, measured slower than MIOpen (5.7 vs 3.5 ms
per 4 frames); was jmodt_amd/ops/rows.py + the JM_DECONV_GEMM branch of train_rows._image_fusion_map.  Needs tools/bin/libjmodt_hip_tools.so
loaded in place of the product library (JM_LIB=tools)."""
import ctypes
from typing import Sequence

import torch
from torch.autograd import Function

from jmodt_amd import _lib as L
from jmodt_amd.ops.rows import _f32, _ptr, _ws

# ---------------------------------------------------------------------------------------------------- deconvolution pyramid
def _cl_ptr(t: torch.Tensor, name: str):
    """raw pointer of a float32 channels-last (B, C, H, W) GPU tensor"""
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == _f32 and t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last)):
        raise RuntimeError(f"{name} must be a float32 channels-last (B, C, H, W) GPU tensor")
    return ctypes.c_void_p(t.data_ptr())


class _DeconvPyramid(Function):
    """cat_i ConvTranspose2d_i(map_i) (kernel == stride k_i, no bias; backbone.py:187-189) as one GEMM per level writing its channel
    slice of the channels-last result directly (csrc/rows_gemm.hip, pixel-shuffled output): apply(ks, H, W, map_1.., wt_1..) with
    map_i (B, C_i, H / k_i, W / k_i) channels-last and wt_i (k_i k_i r_i, C_i) = W_i.permute(2, 3, 1, 0) rows.  No concatenation, no
    library convolution; the backward reads the gradient of the concatenated map in place (dgrad, wgrad GEMMs per level)."""

    @staticmethod
    def forward(ctx, ks, H, W, *tens):
        lib = L.load()
        nl = len(ks)
        maps, wts = tens[:nl], tens[nl:]
        B = maps[0].shape[0]
        rs = [wt.shape[0] // (k * k) for wt, k in zip(wts, ks)]
        ctot = sum(rs)
        de = torch.empty((B, ctot, H, W), dtype=_f32, device=maps[0].device, memory_format=torch.channels_last)
        coff, keep = 0, []
        for mp, wt, k, r in zip(maps, wts, ks, rs):
            if not mp.is_contiguous(memory_format=torch.channels_last):
                mp = mp.contiguous(memory_format=torch.channels_last)
            _, C, h, w = mp.shape
            if h * k != H or w * k != W:
                raise ValueError(f"deconv pyramid: a {h} x {w} map with kernel = stride {k} does not give {H} x {W}")
            wt = wt.contiguous()
            L.check(lib.jm_rows_deconv_forward(B * h * w, C, k, r, h, w, _cl_ptr(mp, "map"), C, L.dev(wt, _f32, "wt"), _ptr(de), ctot, coff,
                                               L.stream_ptr()), "rows_deconv_forward")
            keep += [mp, wt]
            coff += r
        ctx.ks, ctx.rs, ctx.dims = tuple(ks), tuple(rs), (B, ctot, H, W)
        ctx.save_for_backward(*keep)
        return de

    @staticmethod
    def backward(ctx, dde):
        lib = L.load()
        B, ctot, H, W = ctx.dims
        if not dde.is_contiguous(memory_format=torch.channels_last):
            dde = dde.contiguous(memory_format=torch.channels_last)
        saved = ctx.saved_tensors
        nl = len(ctx.ks)
        dmaps, dwts, coff = [], [], 0
        for i, (k, r) in enumerate(zip(ctx.ks, ctx.rs)):
            mp, wt = saved[2 * i], saved[2 * i + 1]
            _, C, h, w = mp.shape
            m = B * h * w
            dm = dw = None
            if ctx.needs_input_grad[3 + i]:
                dm = torch.empty_like(mp)                     # channels-last, like the map
                L.check(lib.jm_rows_deconv_dgrad(m, C, k, r, h, w, _cl_ptr(dde, "dde"), ctot, coff, L.dev(wt, _f32, "wt"), _ptr(dm), C,
                                                 L.stream_ptr()), "rows_deconv_dgrad")
            if ctx.needs_input_grad[3 + nl + i]:
                dw = torch.empty_like(wt)
                nbytes = int(lib.jm_rows_wgrad_workspace_bytes(m, k * k * r, C))
                ws = _ws(nbytes, dde.device) if nbytes else None
                L.check(lib.jm_rows_deconv_wgrad(m, C, k, r, h, w, _cl_ptr(dde, "dde"), ctot, coff, _cl_ptr(mp, "map"), C, _ptr(dw),
                                                 _ptr(ws), nbytes, L.stream_ptr()), "rows_deconv_wgrad")
            dmaps.append(dm)
            dwts.append(dw)
            coff += r
        return (None, None, None, *dmaps, *dwts)


def deconv_pyramid(maps: Sequence[torch.Tensor], weights: Sequence[torch.Tensor], ks: Sequence[int]) -> torch.Tensor:
    """maps[i] (B, C_i, h_i, w_i) channels-last, weights[i] = the ConvTranspose2d weight (C_i, r_i, k_i, k_i) -> the channels-last
    (B, sum r_i, h_i k_i, w_i k_i) concatenation of the transposed convolutions (biases are the caller's: they commute with what follows)"""
    H, W = maps[0].shape[2] * ks[0], maps[0].shape[3] * ks[0]
    wts = [wd.permute(2, 3, 1, 0).reshape(k * k * wd.shape[1], wd.shape[0]) for wd, k in zip(weights, ks)]
    return _DeconvPyramid.apply(tuple(int(k) for k in ks), int(H), int(W), *maps, *wts)


