"""Fused set-abstraction scale: group + SharedMLP (eval-mode BN folded) + max-pool in ONE kernel
(jmodt_amd/csrc/sa_mlp.hip).  No reference counterpart as a function: it replaces the per-scale
body of `_PointnetSAModuleBase.forward` (jmodt/ops/pointnet2/pointnet2_modules.py:46-52)."""
import ctypes
import weakref
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


_shape_cache = weakref.WeakKeyDictionary()    # module -> [(cout, cin)] | None


def _layer_shapes(mlp: nn.Sequential):
    """[(cout, cin)] if every unit is Conv2d(1x1)[+BN]+ReLU in post-activation order, else None.
    Structure only (no tensor math), cached per module object."""
    if mlp in _shape_cache:
        return _shape_cache[mlp]
    shapes = []
    for unit in mlp.children():
        conv = getattr(unit, "conv", None)
        if (not isinstance(conv, nn.Conv2d) or conv.kernel_size != (1, 1) or getattr(unit, "activation", None) is None
                or list(unit._modules)[0] != "conv" or hasattr(unit, "in") or not isinstance(unit.activation, nn.ReLU)):
            shapes = None
            break
        shapes.append((conv.out_channels, conv.in_channels))
    _shape_cache[mlp] = shapes
    return shapes


def can_fuse(mlp: nn.Sequential, npoint: int, nsample: int, training: bool) -> bool:
    if training or nsample not in (16, 32, 64) or (npoint * nsample) % 128:
        return False
    shapes = _layer_shapes(mlp)
    if not shapes or len(shapes) > 4:
        return False
    if len(shapes) == 1 and shapes[0][1] > 128:
        return False
    return all(cout <= 128 for cout, _ in shapes[:-1]) and next(mlp.parameters()).is_cuda


_packed_cache = weakref.WeakKeyDictionary()   # module -> (signature, packed layers)


def _packed_layers(mlp: nn.Sequential, device):
    """[(wp, bp, cout, cin)] in the kernel's device layout (jm_sa_mlp_pack), cached per module until a
    parameter or BatchNorm buffer changes (torch bumps `_version` on every in-place update)."""
    tensors = [t for t in list(mlp.parameters()) + list(mlp.buffers())]
    sig = tuple((t.data_ptr(), t._version) for t in tensors) + (str(device),)
    hit = _packed_cache.get(mlp)
    if hit is not None and hit[0] == sig:
        return hit[1]
    lib = L.load()
    packed = []
    for li, (W, b) in enumerate(fold_shared_mlp(mlp)):
        W = W.to(device=device, dtype=_f32).contiguous()
        b = b.to(device=device, dtype=_f32).contiguous()
        cout, cin = W.shape
        first = 1 if li == 0 else 0
        wp = torch.empty((lib.jm_sa_mlp_packed_weight_elems(cout, cin, first),), dtype=_f32, device=device)
        bp = torch.empty((lib.jm_sa_mlp_packed_bias_elems(cout),), dtype=_f32, device=device)
        L.check(lib.jm_sa_mlp_pack(cout, cin, first, L.dev(W, _f32, "W"), L.dev(b, _f32, "b"), ctypes.c_void_p(wp.data_ptr()),
                                   ctypes.c_void_p(bp.data_ptr()), L.stream_ptr()), "sa_mlp_pack")
        packed.append((wp, bp, cout, cin))
    _packed_cache[mlp] = (sig, packed)
    return packed


@torch.no_grad()
def sa_mlp_fused(xyz: torch.Tensor, new_xyz: torch.Tensor, features: Optional[torch.Tensor], idx: torch.Tensor,
                 mlp: nn.Sequential) -> torch.Tensor:
    """xyz (B,N,3), new_xyz (B,M,3), features (B,C,N) or None, idx (B,M,ns) int32 -> (B, mlp_out, M)"""
    lib = L.load()
    layers = _packed_layers(mlp, xyz.device)
    B, N, _ = xyz.shape
    M, ns = idx.shape[1], idx.shape[2]
    C = 0 if features is None else features.shape[1]
    widths = [3 + C] + [cout for _, _, cout, _ in layers]
    if layers[0][3] != widths[0]:
        raise ValueError(f"SharedMLP expects {layers[0][3]} input channels, got 3 + {C}")
    nl = len(layers)
    out = torch.empty((B, widths[-1], M), dtype=_f32, device=xyz.device)
    feats = features.to(_f32).contiguous() if features is not None else None
    warr = (ctypes.c_void_p * nl)(*[wp.data_ptr() for wp, _, _, _ in layers])
    barr = (ctypes.c_void_p * nl)(*[bp.data_ptr() for _, bp, _, _ in layers])
    widths_c = (ctypes.c_int * (nl + 1))(*widths)
    L.check(lib.jm_sa_mlp_forward(B, N, M, C, ns, L.dev(xyz.contiguous(), _f32, "xyz"),
                                  L.dev(new_xyz.contiguous(), _f32, "new_xyz"),
                                  L.dev(feats, _f32, "features") if feats is not None else None,
                                  L.dev(idx, _i32, "idx"), nl, widths_c, warr, barr,
                                  ctypes.c_void_p(out.data_ptr()), L.stream_ptr()), "sa_mlp_fused")
    return out
