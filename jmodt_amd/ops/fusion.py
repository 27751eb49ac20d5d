"""LI-Fusion point -> image gather on the gfx950 kernel.

Mirror of `feature_gather` (jmodt/detection/modeling/backbone.py:79-89):
    F.grid_sample(feature_map.float(), xy.unsqueeze(1), align_corners=True).squeeze(2)
bilinear, zero padding.  The feature map is consumed through its strides, so a
`torch.channels_last` map (what MIOpen prefers for the image branch's convolutions anyway) is
gathered with 16-byte tap reads and without a layout copy.
"""
import torch
from torch.autograd import Function

from .. import _lib as L

_f32 = torch.float32


def _dense_strides(t: torch.Tensor):
    """a (B,C,H,W) tensor whose memory is a permutation of a dense block (NCHW or channels-last)"""
    if t.is_contiguous() or t.is_contiguous(memory_format=torch.channels_last):
        return t
    return t.contiguous()


class _FeatureGather(Function):
    @staticmethod
    def forward(ctx, feature_map: torch.Tensor, xy: torch.Tensor) -> torch.Tensor:
        fm = _dense_strides(feature_map.float())
        xy = xy.float().contiguous()
        B, C, H, W = fm.shape
        N = xy.shape[1]
        out = torch.empty((B, C, N), dtype=_f32, device=fm.device)
        if not fm.is_cuda:
            raise RuntimeError("feature_gather: feature_map must be a GPU tensor (no CPU path)")
        sb, sc, sh, sw = fm.stride()
        import ctypes
        L.check(L.load().jm_feature_gather(B, C, H, W, N, ctypes.c_void_p(fm.data_ptr()), sb, sc, sh, sw,
                                           L.dev(xy, _f32, "xy"), L.dev(out, _f32, "out"), L.stream_ptr()),
                "feature_gather")
        ctx.save_for_backward(xy)
        ctx.geom = (B, C, H, W, N, fm.is_contiguous())
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (xy,) = ctx.saved_tensors
        B, C, H, W, N, nchw = ctx.geom
        fmt = torch.contiguous_format if nchw else torch.channels_last
        grad_map = torch.empty((B, C, H, W), dtype=_f32, device=grad_out.device, memory_format=fmt).zero_()
        sb, sc, sh, sw = grad_map.stride()
        g = grad_out.contiguous()
        import ctypes
        L.check(L.load().jm_feature_gather_grad(B, C, H, W, N, L.dev(g, _f32, "grad_out"), L.dev(xy, _f32, "xy"),
                                                ctypes.c_void_p(grad_map.data_ptr()), sb, sc, sh, sw, L.stream_ptr()),
                "feature_gather.backward")
        return grad_map, None


def feature_gather(feature_map: torch.Tensor, xy: torch.Tensor) -> torch.Tensor:
    """feature_map (B, C, H, W), xy (B, N, 2) normalised to [-1, 1] -> (B, C, N)"""
    return _FeatureGather.apply(feature_map, xy)
