"""Fused set-abstraction scale: group + SharedMLP (eval-mode BN folded) + max-pool in ONE kernel
(jmodt_amd/csrc/sa_mlp.hip).  No reference counterpart as a function: it replaces the per-scale
body of `_PointnetSAModuleBase.forward` (jmodt/ops/pointnet2/pointnet2_modules.py:46-52)."""
import ctypes
from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from ... import _lib as L

_f32, _i32 = torch.float32, torch.int32


def _pad(v: int, mult: int) -> int:
    return (v + mult - 1) // mult * mult


def fold_shared_mlp(mlp: nn.Sequential) -> Optional[List[Tuple[torch.Tensor, torch.Tensor]]]:
    """[(W (out,in), b (out))] of a SharedMLP of post-activation Conv2d(1x1)[+BN]+ReLU units with the
    BatchNorm running statistics folded in; None when the stack has a shape the kernel does not cover."""
    layers = []
    for unit in mlp.children():
        conv = getattr(unit, "conv", None)
        if not isinstance(conv, nn.Conv2d) or conv.kernel_size != (1, 1) or getattr(unit, "activation", None) is None:
            return None
        if list(unit._modules)[0] != "conv" or hasattr(unit, "in"):   # pre-activation / instance norm: not covered
            return None
        if not isinstance(unit.activation, nn.ReLU):
            return None
        W = conv.weight.detach().to(_f32).view(conv.out_channels, conv.in_channels)
        b = conv.bias.detach().to(_f32) if conv.bias is not None else torch.zeros(conv.out_channels, device=W.device)
        bn_wrap = getattr(unit, "bn", None)
        if bn_wrap is not None:
            bn = bn_wrap.bn
            scale = bn.weight.detach() / torch.sqrt(bn.running_var.detach() + bn.eps)
            W = W * scale[:, None]
            b = (b - bn.running_mean.detach()) * scale + bn.bias.detach()
        layers.append((W, b))
    return layers


def can_fuse(mlp: nn.Sequential, npoint: int, nsample: int, training: bool) -> bool:
    if training or nsample not in (16, 32, 64) or (npoint * nsample) % 128:
        return False
    layers = fold_shared_mlp(mlp)
    if not layers or len(layers) > 4:
        return False
    return all(W.shape[0] <= 128 for W, _ in layers[:-1]) and layers[0][0].is_cuda


@torch.no_grad()
def sa_mlp_fused(xyz: torch.Tensor, new_xyz: torch.Tensor, features: Optional[torch.Tensor], idx: torch.Tensor,
                 mlp: nn.Sequential) -> torch.Tensor:
    """xyz (B,N,3), new_xyz (B,M,3), features (B,C,N) or None, idx (B,M,ns) int32 -> (B, mlp_out, M)"""
    lib = L.load()
    layers = fold_shared_mlp(mlp)
    B, N, _ = xyz.shape
    M, ns = idx.shape[1], idx.shape[2]
    C = 0 if features is None else features.shape[1]
    widths = [3 + C] + [W.shape[0] for W, _ in layers]
    if layers[0][0].shape[1] != widths[0]:
        raise ValueError(f"SharedMLP expects {layers[0][0].shape[1]} input channels, got 3 + {C}")
    nl = len(layers)
    keep, wp, bp = [], [], []
    for l, (W, b) in enumerate(layers):
        kp = _pad(widths[l], 16)
        npad = _pad(widths[l + 1], 128 if l == nl - 1 else 16)
        Wp = torch.zeros((npad, kp), dtype=_f32, device=xyz.device)
        Wp[: W.shape[0], : W.shape[1]] = W
        bpad = torch.zeros((npad,), dtype=_f32, device=xyz.device)
        bpad[: b.shape[0]] = b
        keep += [Wp, bpad]
        wp.append(Wp.data_ptr()); bp.append(bpad.data_ptr())
    out = torch.empty((B, widths[-1], M), dtype=_f32, device=xyz.device)
    feats = features.to(_f32).contiguous() if features is not None else None
    warr = (ctypes.c_void_p * nl)(*wp)
    barr = (ctypes.c_void_p * nl)(*bp)
    widths_c = (ctypes.c_int * (nl + 1))(*widths)
    L.check(lib.jm_sa_mlp_forward(B, N, M, C, ns, L.dev(xyz.contiguous(), _f32, "xyz"),
                                  L.dev(new_xyz.contiguous(), _f32, "new_xyz"),
                                  L.dev(feats, _f32, "features") if feats is not None else None,
                                  L.dev(idx, _i32, "idx"), nl, widths_c, warr, barr,
                                  ctypes.c_void_p(out.data_ptr()), L.stream_ptr()), "sa_mlp_fused")
    return out
