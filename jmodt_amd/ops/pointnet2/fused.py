"""Fused set-abstraction scale: group + SharedMLP (eval-mode BN folded) + max-pool in ONE kernel
(jmodt_amd/csrc/sa_mlp.hip).  No reference counterpart as a function: it replaces the per-scale
body of `_PointnetSAModuleBase.forward` (jmodt/ops/pointnet2/pointnet2_modules.py:46-52)."""
import ctypes
import weakref
from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from ... import _lib as L
from ..._registry import module_tensors, tensor_sig
from ...profile import prof

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


def can_fuse(mlp: nn.Sequential, npoint: int, nsample: int, training: bool, batch: int = 1, n: int = 0,
             group_all: bool = False) -> bool:
    """does one of the two fused kernels (jm_sa_mlp_supported) take this scale?  npoint / nsample as the grouper
    produces them (GroupAll: npoint = 1, nsample = N)"""
    if training:
        return False          # batch statistics: BatchNorm cannot be folded
    shapes = _layer_shapes(mlp)
    if not shapes or not module_tensors(mlp)[0].is_cuda:
        return False
    widths = [shapes[0][1]] + [cout for cout, _ in shapes]
    arr = (ctypes.c_int * len(widths))(*widths)
    return L.load().jm_sa_mlp_supported(int(batch), int(n) if n else max(nsample, 1), int(npoint), widths[0] - 3, int(nsample),
                                        int(group_all), len(shapes), arr) != 0


_packed_cache = weakref.WeakKeyDictionary()   # module -> (signature, packed layers)


def _packed_layers(mlp: nn.Sequential, device):
    """[(wp, bp, cout, cin)] in the kernel's device layout (jm_sa_mlp_pack), cached per module until a
    parameter or BatchNorm buffer changes (torch bumps `_version` on every in-place update)."""
    tensors = module_tensors(mlp)
    sig = tuple(tensor_sig(t) for t in tensors) + (str(device),)
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


CONV1D_STACK = True       # per-point Conv1d chains (FP MLPs, hoisted first SA layer) as one launch (csrc/conv1d_stack.hip)
STACK_MIN_TILES = 128     # below this many 32-point tiles the launch cannot fill the machine: library GEMMs
_stack_cache = weakref.WeakKeyDictionary()    # module -> {operand widths: (folded layers it was packed from, PackedConv1dStack)}
_pre_cache = weakref.WeakKeyDictionary()      # module -> (signature, (W1, b1, packed layers 2..L))
PRE_PROJECT = True    # first layer as per-point / per-centre GEMMs in front of the kernel where that applies


PM_KERNEL = True      # two-layer pre-projected scales on csrc/sa_mlp_pm.hip (point-major u, two MFMA waves per SIMD)


def _pack_k8(W: torch.Tensor, b: torch.Tensor):
    """jm_sa_mlp_pack of a (cout, cin) weight whose columns are first zero padded to a multiple of 16 and permuted within
    every 16-block so that MFMA lane (k-half lk, step kk) reads column 16 kt + 8 lk + kk (csrc/sa_mlp_pm.hip)"""
    lib = L.load()
    cout, cin = W.shape
    kp = (cin + 15) // 16 * 16
    Wz = torch.zeros((cout, kp), dtype=_f32, device=W.device)
    Wz[:, :cin] = W
    j = torch.arange(kp, device=W.device)
    perm = (j // 16) * 16 + 8 * (j % 2) + (j % 16) // 2
    Wq = Wz[:, perm].contiguous()
    wp = torch.empty((lib.jm_sa_mlp_packed_weight_elems(cout, kp, 0),), dtype=_f32, device=W.device)
    bp = torch.empty((lib.jm_sa_mlp_packed_bias_elems(cout),), dtype=_f32, device=W.device)
    L.check(lib.jm_sa_mlp_pack(cout, kp, 0, L.dev(Wq, _f32, "W"), L.dev(b.contiguous(), _f32, "b"), ctypes.c_void_p(wp.data_ptr()),
                               ctypes.c_void_p(bp.data_ptr()), L.stream_ptr()), "sa_mlp_pack")
    return wp, bp


_width_cache = weakref.WeakKeyDictionary()    # module -> (registration epoch, width)


def out_width(mlp: nn.Sequential) -> int:
    """output channels of a SharedMLP (its last convolution's width); the module walk (16 per composed step, 1100 `named_modules`
    frames) is repeated only after a module / parameter registration somewhere in the process (_registry.EPOCH)"""
    from ..._registry import EPOCH
    hit = _width_cache.get(mlp)
    if hit is not None and hit[0] == EPOCH[0]:
        return hit[1]
    w = None
    for m in mlp.modules():
        if isinstance(m, (nn.Conv2d, nn.Conv1d)):
            w = m.out_channels
    if w is None:
        raise ValueError("SharedMLP without a convolution")
    _width_cache[mlp] = (EPOCH[0], int(w))
    return int(w)


def _out_slot(out: Optional[torch.Tensor], B: int, cout: int, M: int, device):
    """(tensor, frame stride in floats): `out` = a (B, cout, M) view whose channel rows are contiguous (a channel slice of a wider
    (B, Ctot, M) tensor: the MSG concatenation written in place), or None for a fresh tensor"""
    if out is None:
        return torch.empty((B, cout, M), dtype=_f32, device=device), 0
    assert out.dtype == _f32 and tuple(out.shape) == (B, cout, M) and out.stride(2) == 1 and (out.stride(1) == M or cout == 1) \
        and (B == 1 or out.stride(0) >= cout * M), (out.shape, out.stride())
    return out, (out.stride(0) if B > 1 else cout * M)


def _pm_layers(mlp: nn.Sequential, device, extra: dict):
    """(w_hidden, b_hidden, w_out, b_out, hidden, cout) of a scale with exactly two layers after the hoisted one, else None"""
    if "pm" not in extra:
        folded = fold_shared_mlp(mlp)
        if len(folded) != 3:
            extra["pm"] = None
        else:
            (W2, b2), (W3, b3) = [(W.to(device=device, dtype=_f32), b.to(device=device, dtype=_f32)) for W, b in folded[1:]]
            extra["pm"] = _pack_k8(W2, b2) + _pack_k8(W3, b3) + (W2.shape[0], W3.shape[0])
    return extra["pm"]


def _sa_mlp_pm(u_pm: torch.Tensor, new_xyz: torch.Tensor, idx: torch.Tensor, w1x: torch.Tensor, pm, out=None, listed: bool = True,
               plan=None) -> torch.Tensor:
    """u_pm (B, N, C) point-major -> (B, cout, M) through jm_sa_mlp_pm_forward (or its listed form: same bits, the rows a group
    of d distinct neighbours executes are 2^max(2, ceil(log2 d)) instead of nsample)"""
    wh, bh, wo, bo, hidden, cout = pm
    B, N, C = u_pm.shape
    M, ns = idx.shape[1], idx.shape[2]
    out, stride = _out_slot(out, B, cout, M, u_pm.device)
    lib = L.load()
    if listed and LISTED and lib.jm_sa_mlp_pm_listed_supported(B, N, M, C, ns, hidden, cout):
        idx = idx.contiguous()
        if plan is None:
            hoisted, prof._hoisted = prof._hoisted, 0        # (the caller's hoisted-layer note belongs to the MLP call, not the plan)
            plan = group_plan(idx, int(lib.jm_sa_mlp_pm_listed_qmin(C, hidden, cout)))
            prof._hoisted = hoisted
        cnt, gl = plan
        L.check(lib.jm_sa_mlp_pm_forward_listed(B, N, M, C, ns, hidden, cout, L.dev(u_pm, _f32, "u"), L.dev(w1x, _f32, "w1x"),
                                                L.dev(new_xyz.contiguous(), _f32, "new_xyz"), L.dev(idx, _i32, "idx"),
                                                L.dev(wh, _f32, "w_hidden"), L.dev(bh, _f32, "b_hidden"), L.dev(wo, _f32, "w_out"),
                                                L.dev(bo, _f32, "b_out"), ctypes.c_void_p(cnt.data_ptr()), ctypes.c_void_p(gl.data_ptr()),
                                                ctypes.c_void_p(out.data_ptr()), stride, L.stream_ptr()), "sa_mlp_pm(listed)")
        ListedStats.last.append((prof._key("sa_mlp_pm_forward_listed"), B * M * ns, ns, cnt))
        del ListedStats.last[:-16]
        return out
    L.check(L.load().jm_sa_mlp_pm_forward_into(B, N, M, C, ns, hidden, cout, L.dev(u_pm, _f32, "u"), L.dev(w1x, _f32, "w1x"),
                                               L.dev(new_xyz.contiguous(), _f32, "new_xyz"), L.dev(idx, _i32, "idx"), L.dev(wh, _f32, "w_hidden"),
                                               L.dev(bh, _f32, "b_hidden"), L.dev(wo, _f32, "w_out"), L.dev(bo, _f32, "b_out"),
                                               ctypes.c_void_p(out.data_ptr()), stride, L.stream_ptr()), "sa_mlp_pm")
    return out


DEDUPE = True         # RCNN scales: skip (centre, sample) rows that are exact copies (csrc/sa_dedupe.hip), bit-identical output


DEDUPE_LISTED = True  # ... and run the compacted segments through the LISTED kernel (2^q rows per segment of d entries, not 16)


class DedupeStats:
    """device-side record of the last duplicate-compacted scales: [(name, dense rows, counters tensor, class counts or None)]"""
    last = []

    @staticmethod
    def rows_executed(entry) -> int:
        """rows the MFMA kernel ran for one recorded scale (synchronises): 128 per tile of the 16-row form, or groups << q"""
        _, _, counters, cls = entry
        if cls is None:
            return int(counters[1].item()) * 128
        c = cls[:8].tolist()
        return sum(int(c[q]) << q for q in range(8))


def dedupe_applies(mlp: nn.Sequential, device, R: int, n: int, H1: int, npoint: int, nsample: int) -> bool:
    """will sa_scale_pm_dedupe take this scale?  (callers that skip work for non-canonical rows decide on this)"""
    lib = L.load()
    W1, b1, w1x, packed, extra = _pre_layers(mlp, device)
    pm = _pm_layers(mlp, device, extra)
    cap = int(lib.jm_sa_dedupe_capacity(R, npoint, nsample))
    return bool(DEDUPE and pm is not None and n <= 2048 and npoint <= 256 and R * n < 2 ** 31
                and lib.jm_sa_mlp_pm_supported(1, R * n, cap, H1, 16, pm[4], pm[5]))


@torch.no_grad()
def sa_scale_pm_dedupe(xyz: torch.Tensor, u_pm: torch.Tensor, mlp: nn.Sequential, npoint: int, radius: float, nsample: int,
                       canon: torch.Tensor, name: str = ""):
    """one QueryAndGroup + SharedMLP + max-pool scale on point sets with known exact copies, pre-projected form:
    xyz (R, n, 3), u_pm (R, n, H1) point-major hoisted first layer, canon (R, n) int32 (canon[k] = first point of which point
    k is an exact copy) -> (new_xyz (R, npoint, 3), features (R, mlp_out, npoint), rep (R, npoint) int32 = the next level's
    canon), or None when the scale does not run on sa_mlp_pm_kernel.
    Sampling and neighbour search are the ordinary ones (their outputs are what the reference computes); only the rows the
    MFMA kernel executes are compacted: distinct canonical neighbours of distinct canonical centres, in segments of 16."""
    from . import pointnet2_utils
    lib = L.load()
    R, n, H1 = u_pm.shape
    dev = xyz.device
    W1, b1, w1x, packed, extra = _pre_layers(mlp, dev)
    pm = _pm_layers(mlp, dev, extra)
    cap = int(lib.jm_sa_dedupe_capacity(R, npoint, nsample))
    if (pm is None or n > 2048 or npoint > 256 or R * n >= 2 ** 31 or
            not lib.jm_sa_mlp_pm_supported(1, R * n, cap, H1, 16, pm[4], pm[5])):
        return None
    wh, bh, wo, bo, hidden, cout = pm
    fps_idx, new_xyz = pointnet2_utils.farthest_point_sample_xyz(xyz, npoint)
    nb = pointnet2_utils.ball_query(radius, nsample, xyz, new_xyz)
    rep = torch.empty((R, npoint), dtype=_i32, device=dev)
    seg_start = torch.empty((R, npoint), dtype=_i32, device=dev)
    seg_cnt = torch.empty((R, npoint), dtype=_i32, device=dev)
    vidx = torch.empty((cap, 16), dtype=_i32, device=dev)
    vxyz = torch.empty((cap, 3), dtype=_f32, device=dev)
    counters = torch.empty((4,), dtype=_i32, device=dev)
    L.check(lib.jm_sa_dedupe_plan(R, n, npoint, nsample, L.dev(canon, _i32, "canon"), L.dev(fps_idx, _i32, "fps_idx"),
                                  L.dev(nb, _i32, "nb"), L.dev(new_xyz, _f32, "new_xyz"), L.dev(rep, _i32, "rep"),
                                  L.dev(seg_start, _i32, "seg_start"), L.dev(seg_cnt, _i32, "seg_cnt"), L.dev(vidx, _i32, "vidx"),
                                  L.dev(vxyz, _f32, "vxyz"), L.dev(counters, _i32, "counters"), L.stream_ptr()), "sa_dedupe_plan")
    outv = torch.empty((cout, cap), dtype=_f32, device=dev)
    prof.hoisted_flops(0)
    cls = None
    if LISTED and DEDUPE_LISTED and lib.jm_sa_mlp_pm_listed_supported(1, R * n, cap, H1, 16, hidden, cout):
        # the virtual centres' lists ARE in ball-query form (distinct entries, padded with the first): the listed kernel runs a
        # segment of d entries on 2^max(qmin, ceil(log2 d)) rows instead of 16 — most segments of a sparse RoI hold 1..4 rows
        cls, gl = group_plan(vidx.view(1, cap, 16), int(lib.jm_sa_mlp_pm_listed_qmin(H1, hidden, cout)), counters[0:1])
        L.check(lib.jm_sa_mlp_pm_forward_listed(1, R * n, cap, H1, 16, hidden, cout, L.dev(u_pm.contiguous(), _f32, "u"), L.dev(w1x, _f32, "w1x"),
                                                L.dev(vxyz, _f32, "vxyz"), L.dev(vidx, _i32, "vidx"), L.dev(wh, _f32, "w_hidden"),
                                                L.dev(bh, _f32, "b_hidden"), L.dev(wo, _f32, "w_out"), L.dev(bo, _f32, "b_out"),
                                                ctypes.c_void_p(cls.data_ptr()), ctypes.c_void_p(gl.data_ptr()),
                                                ctypes.c_void_p(outv.data_ptr()), 0, L.stream_ptr()), "sa_mlp_pm(dedupe, listed)")
    else:
        L.check(lib.jm_sa_mlp_pm_forward_dyn(R * n, cap, H1, 16, hidden, cout, L.dev(u_pm.contiguous(), _f32, "u"), L.dev(w1x, _f32, "w1x"),
                                             L.dev(vxyz, _f32, "vxyz"), L.dev(vidx, _i32, "vidx"), L.dev(wh, _f32, "w_hidden"),
                                             L.dev(bh, _f32, "b_hidden"), L.dev(wo, _f32, "w_out"), L.dev(bo, _f32, "b_out"),
                                             ctypes.c_void_p(outv.data_ptr()), ctypes.c_void_p(counters.data_ptr() + 4), L.stream_ptr()),
                "sa_mlp_pm(dedupe)")
    out = torch.empty((R, cout, npoint), dtype=_f32, device=dev)
    L.check(lib.jm_sa_dedupe_combine(R, npoint, cout, cap, L.dev(outv, _f32, "outv"), L.dev(rep, _i32, "rep"),
                                     L.dev(seg_start, _i32, "seg_start"), L.dev(seg_cnt, _i32, "seg_cnt"),
                                     ctypes.c_void_p(out.data_ptr()), L.stream_ptr()), "sa_dedupe_combine")
    DedupeStats.last.append((name, R * npoint * nsample, counters, cls))
    return new_xyz, out, rep


@torch.no_grad()
def canon_from_count(count: torch.Tensor, n: int) -> torch.Tensor:
    """(R,) distinct points per cyclically padded set (roipool3d) -> canon (R, n) int32, canon[k] = k % max(count, 1)"""
    count = count.reshape(-1).to(_i32).contiguous()
    canon = torch.empty((count.shape[0], n), dtype=_i32, device=count.device)
    L.check(L.load().jm_sa_dedupe_canon_from_cnt(count.shape[0], n, L.dev(count, _i32, "count"), L.dev(canon, _i32, "canon"),
                                                 L.stream_ptr()), "sa_dedupe_canon_from_cnt")
    return canon


@torch.no_grad()
def hoisted_u_point_major(xyz: torch.Tensor, features: torch.Tensor, mlp: nn.Sequential) -> Optional[torch.Tensor]:
    """u = W1 [xyz | f] + b1 per point, point-major (B, N, H1): the hoisted first layer of a pre-projected scale as ONE launch
    (csrc/conv1d_stack.hip), or None where that kernel does not take the shape"""
    W1, b1, w1x, packed, extra = _pre_layers(mlp, xyz.device)
    B, N, _ = xyz.shape
    if not (CONV1D_STACK and N % 32 == 0):
        return None
    st = extra.get("u_stack")
    if st is None:
        from ..conv1d import PackedConv1dStack
        st = extra["u_stack"] = PackedConv1dStack([(torch.cat([W1[:, 3:], W1[:, :3]], dim=1), b1, False)], W1.shape[1] - 3, 3, True)
    if not st.supported(B, N):
        return None
    return st(features.to(_f32), xyz, point_major=True)


def pm_plan(mlp: nn.Sequential, device, B: int, N: int, M: int, ns: int):
    """the point-major kernel's packed layers when this scale runs on it (callers then produce u as (B, N, C)), else None"""
    if not PM_KERNEL:
        return None
    W1, b1, w1x, packed, extra = _pre_layers(mlp, device)
    pm = _pm_layers(mlp, device, extra)
    if pm is None or not L.load().jm_sa_mlp_pm_supported(B, N, M, W1.shape[0], ns, pm[4], pm[5]):
        return None
    return pm


def _pre_layers(mlp: nn.Sequential, device):
    """the pre-projected form's operands: (W1 (H1, 3 + C) folded, b1, W1x (H1, 4), [(wp, bp, cout, cin)] of layers 2..L)"""
    tensors = module_tensors(mlp)
    sig = tuple(tensor_sig(t) for t in tensors) + (str(device), "pre")
    hit = _pre_cache.get(mlp)
    if hit is not None and hit[0] == sig:
        return hit[1]
    lib = L.load()
    folded = fold_shared_mlp(mlp)
    W1, b1 = folded[0][0].to(device=device, dtype=_f32).contiguous(), folded[0][1].to(device=device, dtype=_f32).contiguous()
    packed = []
    for W, b in folded[1:]:
        W = W.to(device=device, dtype=_f32).contiguous()
        b = b.to(device=device, dtype=_f32).contiguous()
        cout, cin = W.shape
        wp = torch.empty((lib.jm_sa_mlp_packed_weight_elems(cout, cin, 0),), dtype=_f32, device=device)
        bp = torch.empty((lib.jm_sa_mlp_packed_bias_elems(cout),), dtype=_f32, device=device)
        L.check(lib.jm_sa_mlp_pack(cout, cin, 0, L.dev(W, _f32, "W"), L.dev(b, _f32, "b"), ctypes.c_void_p(wp.data_ptr()),
                                   ctypes.c_void_p(bp.data_ptr()), L.stream_ptr()), "sa_mlp_pack")
        packed.append((wp, bp, cout, cin))
    w1x = torch.zeros((W1.shape[0], 4), dtype=_f32, device=device)
    w1x[:, :3] = W1[:, :3]
    val = (W1, b1, w1x, packed, {})
    _pre_cache[mlp] = (sig, val)
    return val


def _can_pre_project(mlp: nn.Sequential, features, idx, M: int, ns: int) -> bool:
    shapes = _layer_shapes(mlp)
    if not PRE_PROJECT or features is None or idx is None or not shapes or len(shapes) < 3 or len(shapes) > 4:
        return False
    h1 = shapes[0][0]
    return (h1 % 16 == 0 and h1 <= 128 and all(c <= 128 for c, _ in shapes[:-1]) and ns in (16, 32, 64)
            and (M * ns) % 128 == 0)


@torch.no_grad()
def sa_mlp_pre_from_u(u: torch.Tensor, new_xyz: torch.Tensor, idx: torch.Tensor, mlp: nn.Sequential,
                      point_major: bool = False) -> torch.Tensor:
    """layers 2..L + max-pool of a set-abstraction scale whose hoisted first layer u = W1 [xyz | f] + b1 was computed by
    the producer of the features (ops/rcnn_lift.py): u (B, H1, N), or (B, N, H1) with point_major (the layout
    `pm_plan` asks for) -> (B, mlp_out, M)"""
    lib = L.load()
    W1, b1, w1x, packed, _ = _pre_layers(mlp, u.device)
    M, ns = idx.shape[1], idx.shape[2]
    if point_major:
        B, N, H1 = u.shape
        pm = pm_plan(mlp, u.device, B, N, M, ns)
        if pm is not None:
            prof.hoisted_flops(2 * B * M * ns * H1 * W1.shape[1])
            return _sa_mlp_pm(u.contiguous(), new_xyz, idx, w1x, pm)
        u = u.transpose(1, 2).contiguous()
    B, H1, N = u.shape
    widths = [H1] + [cout for _, _, cout, _ in packed]
    nl = len(packed)
    prof.hoisted_flops(2 * B * M * ns * H1 * W1.shape[1])     # the first layer's per-row work this form avoids
    out = torch.empty((B, widths[-1], M), dtype=_f32, device=u.device)
    warr = (ctypes.c_void_p * nl)(*[wp.data_ptr() for wp, _, _, _ in packed])
    barr = (ctypes.c_void_p * nl)(*[bp.data_ptr() for _, bp, _, _ in packed])
    widths_c = (ctypes.c_int * (nl + 1))(*widths)
    L.check(lib.jm_sa_mlp_forward_pre(B, N, M, H1, ns, L.dev(u.contiguous(), _f32, "u"), L.dev(w1x, _f32, "w1x"),
                                      L.dev(new_xyz.contiguous(), _f32, "new_xyz"), L.dev(idx, _i32, "idx"), nl, widths_c, warr,
                                      barr, ctypes.c_void_p(out.data_ptr()), L.stream_ptr()), "sa_mlp_fused(pre)")
    return out


def hoistable_first_layer(mlp: nn.Sequential, npoint: int, nsample: int, device):
    """(W1 (H1, 3 + C), b1) when the scale qualifies for the pre-projected kernel, else None"""
    shapes = _layer_shapes(mlp)
    if (not PRE_PROJECT or not shapes or len(shapes) < 3 or len(shapes) > 4 or shapes[0][0] % 16 or shapes[0][0] > 128
            or any(c > 128 for c, _ in shapes[:-1]) or nsample not in (16, 32, 64) or (npoint * nsample) % 128):
        return None
    W1, b1, _, _, _ = _pre_layers(mlp, device)
    return W1, b1


@torch.no_grad()
def _sa_mlp_fused_pre(xyz, new_xyz, features, idx, mlp, out=None, listed=True, plan=None):
    """QueryAndGroup + SharedMLP + max-pool with the first layer hoisted in front of the gather: W1 [xyz_j - c_i | f_j] + b1
    = u_j - W1x c_i with u = W1 [xyz | f] + b1 per point (two small batched GEMMs accumulating into one tensor); the
    kernel forms relu(u_j - W1x c_i) while gathering and runs layers 2..L (jm_sa_mlp_forward_pre)"""
    lib = L.load()
    W1, b1, w1x, packed, extra = _pre_layers(mlp, xyz.device)
    B, N, _ = xyz.shape
    M, ns = idx.shape[1], idx.shape[2]
    H1 = W1.shape[0]
    feats = features.to(_f32)
    u = None
    pm = pm_plan(mlp, xyz.device, B, N, M, ns)
    if CONV1D_STACK and N % 32 == 0:
        st = extra.get("u_stack")
        if st is None:
            from ..conv1d import PackedConv1dStack
            st = extra["u_stack"] = PackedConv1dStack([(torch.cat([W1[:, 3:], W1[:, :3]], dim=1), b1, False)], W1.shape[1] - 3, 3, True)
        if st.supported(B, N):
            u = st(feats, xyz, point_major=pm is not None)                            # (B, H1, N) or (B, N, H1), one launch
            if pm is not None:
                prof.hoisted_flops(2 * B * M * ns * H1 * W1.shape[1])
                return _sa_mlp_pm(u, new_xyz, idx, w1x, pm, out, listed, plan)
    if u is None:
        u = torch.baddbmm(b1[None, :, None], W1[:, 3:].expand(B, -1, -1), feats)
        u = u.baddbmm_(W1[:, :3].expand(B, -1, -1), xyz.transpose(1, 2))
        if pm is not None:
            prof.hoisted_flops(2 * B * M * ns * H1 * W1.shape[1])
            return _sa_mlp_pm(u.transpose(1, 2).contiguous(), new_xyz, idx, w1x, pm, out, listed, plan)
    widths = [H1] + [cout for _, _, cout, _ in packed]
    nl = len(packed)
    prof.hoisted_flops(2 * B * M * ns * H1 * W1.shape[1])     # the first layer's per-row work this form avoids
    out, stride = _out_slot(out, B, widths[-1], M, xyz.device)
    warr = (ctypes.c_void_p * nl)(*[wp.data_ptr() for wp, _, _, _ in packed])
    barr = (ctypes.c_void_p * nl)(*[bp.data_ptr() for _, bp, _, _ in packed])
    widths_c = (ctypes.c_int * (nl + 1))(*widths)
    L.check(lib.jm_sa_mlp_forward_pre_into(B, N, M, H1, ns, L.dev(u.contiguous(), _f32, "u"), L.dev(w1x, _f32, "w1x"),
                                           L.dev(new_xyz.contiguous(), _f32, "new_xyz"), L.dev(idx, _i32, "idx"), nl, widths_c, warr,
                                           barr, ctypes.c_void_p(out.data_ptr()), stride, L.stream_ptr()), "sa_mlp_fused(pre)")
    return out


LISTED = True         # duplicate-aware form of the RPN scales (csrc/sa_groups.hip): rows executed 2^ceil(log2 d) per group of d
                      # distinct neighbours instead of nsample; bit-identical output


def listed_kind(mlp: nn.Sequential, features, idx: torch.Tensor, B: int, N: int) -> int:
    """which kernel takes the listed form of this scale (0: none — the dense entry is used)"""
    if not LISTED or idx is None:
        return 0
    M, ns = idx.shape[1], idx.shape[2]
    if _can_pre_project(mlp, features, idx, M, ns):
        return 0
    shapes = _layer_shapes(mlp)
    if not shapes:
        return 0
    widths = [shapes[0][1]] + [cout for cout, _ in shapes]
    arr = (ctypes.c_int * len(widths))(*widths)
    return int(L.load().jm_sa_mlp_listed_supported(int(B), int(N), int(M), widths[0] - 3, int(ns), len(shapes), arr))


@torch.no_grad()
def group_plan(idx: torch.Tensor, qmin: int = 0, groups_dev: Optional[torch.Tensor] = None):
    """idx (B, M, ns) int32 neighbour lists -> the listed form's plan (cls_count (8,) int32 = groups per class of 2^q rows,
    glist = the classes' group ids), both in device memory; groups_dev: (1,) int32 on the device = how many of the B * M groups
    are valid (the rest of idx is never read)"""
    lib = L.load()
    B, M, ns = idx.shape
    buf = torch.empty((8 + int(lib.jm_sa_group_list_elems(B * M, ns)),), dtype=_i32, device=idx.device)
    cnt, gl = buf[:8], buf[8:]
    if groups_dev is not None:
        L.check(lib.jm_sa_group_plan_dev(B * M, ns, L.dev(idx, _i32, "idx"), int(qmin), ctypes.c_void_p(groups_dev.data_ptr()),
                                         ctypes.c_void_p(cnt.data_ptr()), ctypes.c_void_p(gl.data_ptr()), L.stream_ptr()), "sa_group_plan_dev")
    else:
        L.check(lib.jm_sa_group_plan(B * M, ns, L.dev(idx, _i32, "idx"), int(qmin), ctypes.c_void_p(cnt.data_ptr()),
                                     ctypes.c_void_p(gl.data_ptr()), L.stream_ptr()), "sa_group_plan")
    return cnt, gl


@torch.no_grad()
def group_plan_dual(idx0: torch.Tensor, qmin0: int, idx1: torch.Tensor, qmin1: int):
    """the plans of the two scales of a multi-scale level (same centres) from ONE launch: ((cnt0, glist0), (cnt1, glist1))"""
    lib = L.load()
    B, M, ns0 = idx0.shape
    ns1 = idx1.shape[2]
    assert idx1.shape[:2] == (B, M)
    n0, n1 = int(lib.jm_sa_group_list_elems(B * M, ns0)), int(lib.jm_sa_group_list_elems(B * M, ns1))
    buf = torch.empty((16 + n0 + n1,), dtype=_i32, device=idx0.device)
    gl0, gl1 = buf[16:16 + n0], buf[16 + n0:]
    L.check(lib.jm_sa_group_plan_dual(B * M, ns0, L.dev(idx0, _i32, "idx0"), int(qmin0), ns1, L.dev(idx1, _i32, "idx1"), int(qmin1),
                                      ctypes.c_void_p(buf.data_ptr()), ctypes.c_void_p(gl0.data_ptr()), ctypes.c_void_p(gl1.data_ptr()),
                                      L.stream_ptr()), "sa_group_plan_dual")
    return (buf[:8], gl0), (buf[8:16], gl1)


def listed_qmin(mlp: nn.Sequential, features, idx: torch.Tensor, B: int, N: int) -> int:
    """the smallest class (log2 rows) of the kernel that will take this scale in the listed form, -1 when none will"""
    if not LISTED or idx is None or features is not None and not features.is_cuda:
        return -1
    lib = L.load()
    M, ns = idx.shape[1], idx.shape[2]
    if _can_pre_project(mlp, features, idx, M, ns):
        pm = pm_plan(mlp, idx.device, B, N, M, ns)
        if pm is None:
            return -1
        W1 = _pre_layers(mlp, idx.device)[0]
        return (int(lib.jm_sa_mlp_pm_listed_qmin(W1.shape[0], pm[4], pm[5]))
                if lib.jm_sa_mlp_pm_listed_supported(B, N, M, W1.shape[0], ns, pm[4], pm[5]) else -1)
    kind = listed_kind(mlp, features, idx, B, N)
    return int(lib.jm_sa_mlp_listed_qmin(kind)) if kind else -1


class ListedStats:
    """device-side record of the last listed scales: [(name, dense rows, plan tensor)] (bench.py reads the class counts back)"""
    last = []


@torch.no_grad()
def sa_mlp_fused(xyz: torch.Tensor, new_xyz: Optional[torch.Tensor], features: Optional[torch.Tensor],
                 idx: Optional[torch.Tensor], mlp: nn.Sequential, out: Optional[torch.Tensor] = None, listed: bool = True,
                 plan=None) -> torch.Tensor:
    """xyz (B,N,3), new_xyz (B,M,3), features (B,C,N) or None, idx (B,M,ns) int32 -> (B, mlp_out, M);
    idx = new_xyz = None: GroupAll (one group of all N points per frame, xyz not re-centred) -> (B, mlp_out, 1).
    out: optional (B, mlp_out, M) view to write into — a channel slice of a wider tensor (the MSG concatenation in place).
    listed: take the duplicate-aware form where a kernel has one (same bits, fewer rows); plan: its (cls_count, glist) when the
    caller planned already (group_plan_dual for the two scales of a level, with this scale's listed_qmin)"""
    lib = L.load()
    if idx is not None and _can_pre_project(mlp, features, idx, idx.shape[1], idx.shape[2]):
        return _sa_mlp_fused_pre(xyz, new_xyz, features, idx, mlp, out, listed, plan)
    layers = _packed_layers(mlp, xyz.device)
    B, N, _ = xyz.shape
    M, ns = (idx.shape[1], idx.shape[2]) if idx is not None else (1, N)
    C = 0 if features is None else features.shape[1]
    widths = [3 + C] + [cout for _, _, cout, _ in layers]
    if layers[0][3] != widths[0]:
        raise ValueError(f"SharedMLP expects {layers[0][3]} input channels, got 3 + {C}")
    nl = len(layers)
    out, stride = _out_slot(out, B, widths[-1], M, xyz.device)
    feats = features.to(_f32).contiguous() if features is not None else None
    warr = (ctypes.c_void_p * nl)(*[wp.data_ptr() for wp, _, _, _ in layers])
    barr = (ctypes.c_void_p * nl)(*[bp.data_ptr() for _, bp, _, _ in layers])
    widths_c = (ctypes.c_int * (nl + 1))(*widths)
    kind = listed_kind(mlp, features, idx, B, N) if listed else 0
    if kind:
        idx = idx.contiguous()
        cnt, gl = plan if plan is not None else group_plan(idx, int(lib.jm_sa_mlp_listed_qmin(kind)))
        L.check(lib.jm_sa_mlp_forward_listed(B, N, M, C, ns, L.dev(xyz.contiguous(), _f32, "xyz"), L.dev(new_xyz.contiguous(), _f32, "new_xyz"),
                                             L.dev(feats, _f32, "features") if feats is not None else None, L.dev(idx, _i32, "idx"),
                                             nl, widths_c, warr, barr, ctypes.c_void_p(cnt.data_ptr()), ctypes.c_void_p(gl.data_ptr()),
                                             ctypes.c_void_p(out.data_ptr()), stride, L.stream_ptr()), "sa_mlp_fused(listed)")
        ListedStats.last.append((prof._key("sa_mlp_forward_listed"), B * M * ns, ns, cnt))
        del ListedStats.last[:-16]
        return out
    L.check(lib.jm_sa_mlp_forward_into(B, N, M, C, ns, L.dev(xyz.contiguous(), _f32, "xyz"),
                                       L.dev(new_xyz.contiguous(), _f32, "new_xyz") if new_xyz is not None else None,
                                       L.dev(feats, _f32, "features") if feats is not None else None,
                                       L.dev(idx, _i32, "idx") if idx is not None else None, nl, widths_c, warr, barr,
                                       ctypes.c_void_p(out.data_ptr()), stride, L.stream_ptr()), "sa_mlp_fused")
    return out


_folded_cache = weakref.WeakKeyDictionary()   # module -> (signature, [(W, b)])


def _folded_layers(mlp: nn.Sequential):
    """[(W (out, in), b (out))] with eval-mode BatchNorm folded, cached until a parameter / buffer changes"""
    tensors = module_tensors(mlp)
    sig = tuple(tensor_sig(t) for t in tensors)
    hit = _folded_cache.get(mlp)
    if hit is not None and hit[0] == sig:
        return hit[1]
    layers = fold_shared_mlp(mlp)
    if layers is not None:
        layers = [(W.contiguous(), b.contiguous()) for W, b in layers]
    _folded_cache[mlp] = (sig, layers)
    return layers


@torch.no_grad()
def shared_mlp_points(mlp: nn.Sequential, parts) -> Optional[torch.Tensor]:
    """eval-mode SharedMLP on per-point features: parts = [(B, C_i, n), ...] stands for their channel concatenation
    (never materialised: the first layer's weight is split column-wise and the partial products accumulate in one
    output).  Each layer is ONE batched GEMM with the folded BatchNorm bias + one in-place ReLU instead of
    conv2d(1x1) + BatchNorm + ReLU on a (B, C, n, 1) tensor.  None when the stack has a shape folding does not cover."""
    layers = _folded_layers(mlp)
    if layers is None or sum(p.shape[1] for p in parts) != layers[0][0].shape[1]:
        return None
    B = parts[0].shape[0]
    n = parts[0].shape[2]
    if CONV1D_STACK and parts[0].is_cuda and len(parts) <= 2 and len(layers) <= 3 and B * (n // 32) >= STACK_MIN_TILES:
        # the whole stack as one launch on 32-point tiles (csrc/conv1d_stack.hip)
        per_mlp = _stack_cache.setdefault(mlp, {})
        key = tuple(p.shape[1] for p in parts)
        st = per_mlp.get(key)
        if st is None or st[0] is not layers:
            from ..conv1d import PackedConv1dStack
            st = per_mlp[key] = (layers, PackedConv1dStack(
                [(W, b, True) for W, b in layers], parts[0].shape[1], parts[1].shape[1] if len(parts) == 2 else 0))
        if st[1].supported(B, n):
            return st[1](*parts)
    if parts[0].is_cuda and len(parts) <= 2 and all(p.dtype == torch.float32 for p in parts):
        # few points (feature propagation level 4: 8 x 256): one launch of independent waves per layer (csrc/points_gemm.hip)
        from ..conv1d import points_linear, points_linear_supported
        k1, k2 = parts[0].shape[1], (parts[1].shape[1] if len(parts) == 2 else 0)
        if points_linear_supported(B, n, k1, k2, layers[0][0].shape[0]) and all(W.shape[1] % 4 == 0 for W, _ in layers):
            x = points_linear(parts[0].contiguous(), layers[0][0], layers[0][1], 1, x2=parts[1].contiguous() if k2 else None)
            for W, b in layers[1:]:
                x = points_linear(x, W, b, 1)
            return x
    W, b = layers[0]
    x, off = None, 0
    for part in parts:
        c = part.shape[1]
        x = torch.baddbmm(b[None, :, None] if x is None else x, W[:, off:off + c].expand(B, -1, -1), part)
        off += c
    x = torch.relu_(x)
    for W, b in layers[1:]:
        x = torch.relu_(torch.baddbmm(b[None, :, None], W.expand(B, -1, -1), x))
    return x
