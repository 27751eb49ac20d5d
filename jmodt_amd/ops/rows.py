"""The training path's operators on ROW-major activations (csrc/rows_gemm.hip, csrc/rows_ops.hip): forward AND backward of the
dense-layer chains, the set-abstraction scale, three_interpolate and the LI-Fusion gather as torch.autograd Functions whose two
directions are hand-written HIP kernels.

What the reference does with torch autograd over (B, C, npoint, nsample) tensors — pytorch_utils.py:6-33 (SharedMLP),
pointnet2_modules.py:46-61 (set abstraction) / :139-153 (feature propagation), pointnet2_utils.py:105-150,156-197 (interpolation /
grouping backward), backbone.py:35-89 (LI-Fusion) — happens here on (rows, channels) tensors: rows = points, or the DISTINCT
(centre, neighbour) pairs of the ball-query groups.  BatchNorm enters in eval mode, folded into the neighbouring weight by the
caller (differentiably: the fold is plain torch arithmetic on the parameters, so d(loss)/d(gamma, beta, W) follow from the folded
weight's gradient by autograd).  No CPU path: every entry raises on non-GPU tensors (jmodt_amd._lib.dev).
"""
import ctypes
from typing import List, Optional, Sequence, Tuple

import torch
from torch.autograd import Function

from .. import _lib as L

_f32, _i32 = torch.float32, torch.int32


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _rows(t: torch.Tensor, name: str) -> Tuple[ctypes.c_void_p, int]:
    """(pointer, leading dimension) of a 2-D float32 row tensor: unit column stride, 16-byte aligned rows"""
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name} must be a GPU tensor (jmodt_amd has no CPU path)")
    if t.dtype != _f32 or t.dim() != 2:
        raise TypeError(f"{name} must be a 2-D float32 tensor, got {t.dtype} {tuple(t.shape)}")
    if t.shape[1] > 1 and t.stride(1) != 1:
        raise RuntimeError(f"{name} must have unit column stride")
    ld = t.stride(0) if t.shape[0] > 1 else max(t.shape[1], t.stride(0))
    if t.data_ptr() % 16 or ld % 4:
        raise RuntimeError(f"{name}: rows must be 16-byte aligned (ld {ld}, offset {t.data_ptr() % 16})")
    return ctypes.c_void_p(t.data_ptr()), int(ld)


def _ws(nbytes: int, device) -> torch.Tensor:
    return torch.empty(((int(nbytes) + 3) // 4,), dtype=_f32, device=device)


# ---------------------------------------------------------------------------------------------------- dense layers on rows
def linear_forward(x1: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, act: int = 0, x2: Optional[torch.Tensor] = None,
                   rowscale: Optional[torch.Tensor] = None, m_dev: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y (M, N) = act([x1 | x2] w^T + bias) (* rowscale per row); act 0 none, 1 ReLU, 2 tanh"""
    M, k1 = x1.shape
    k2 = 0 if x2 is None else x2.shape[1]
    n = w.shape[0]
    if w.shape[1] != k1 + k2:
        raise ValueError(f"weight of {w.shape[1]} input channels for operands of {k1} + {k2}")
    y = torch.empty((M, n), dtype=_f32, device=x1.device)
    px1, ld1 = _rows(x1, "x1")
    px2, ld2 = _rows(x2, "x2") if x2 is not None else (None, 0)
    pw, ldw = _rows(w, "w")
    L.check(L.load().jm_rows_linear_forward(M, _ptr(m_dev), k1, k2, n, px1, ld1, px2, ld2, pw, ldw, _ptr(bias), int(act), _ptr(rowscale),
                                            _ptr(y), n, L.stream_ptr()), "rows_linear_forward")
    return y


def linear_dgrad(dy: torch.Tensor, w: torch.Tensor, k0: int, k: int, mask: Optional[torch.Tensor] = None, m_dev: Optional[torch.Tensor] = None,
                 accumulate_into: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dx (M, k) = (dy (M, N) w[:, k0 : k0 + k]) .* (mask > 0) [+ accumulate_into, in place]"""
    M, n = dy.shape
    dx = accumulate_into if accumulate_into is not None else torch.empty((M, k), dtype=_f32, device=dy.device)
    pdy, lddy = _rows(dy, "dy")
    pw, ldw = _rows(w, "w")
    pm, ldm = _rows(mask, "mask") if mask is not None else (None, 0)
    pdx, lddx = _rows(dx, "dx")
    if k0 % 4:
        raise ValueError("column offset of the second operand must be a multiple of 4")
    L.check(L.load().jm_rows_linear_dgrad(M, _ptr(m_dev), n, k, pdy, lddy, ctypes.c_void_p(w.data_ptr() + 4 * k0), ldw, pm, ldm,
                                          1 if accumulate_into is not None else 0, pdx, lddx, L.stream_ptr()), "rows_linear_dgrad")
    return dx


def linear_wgrad(dy: torch.Tensor, xs: Sequence[torch.Tensor], want_bias: bool = True, m_dev: Optional[torch.Tensor] = None):
    """(dw (N, sum k_i) = dy^T [x_1 | x_2 | ...], dbias (N) or None)"""
    lib = L.load()
    M, n = dy.shape
    ktot = sum(x.shape[1] for x in xs)
    dw = torch.empty((n, ktot), dtype=_f32, device=dy.device)
    db = torch.empty((n,), dtype=_f32, device=dy.device) if want_bias else None
    pdy, lddy = _rows(dy, "dy")
    off = 0
    for i, x in enumerate(xs):
        k = x.shape[1]
        px, ldx = _rows(x, "x")
        nbytes = int(lib.jm_rows_wgrad_workspace_bytes(M, n, k))
        ws = _ws(nbytes, dy.device)
        L.check(lib.jm_rows_linear_wgrad(M, _ptr(m_dev), n, k, pdy, lddy, px, ldx, ctypes.c_void_p(dw.data_ptr() + 4 * off), ktot,
                                         _ptr(db) if (want_bias and i == 0) else None, 0, _ptr(ws), nbytes, L.stream_ptr()), "rows_linear_wgrad")
        off += k
    return dw, db


def colsum(x: torch.Tensor, m_dev: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = L.load()
    M, n = x.shape
    out = torch.empty((n,), dtype=_f32, device=x.device)
    nbytes = int(lib.jm_rows_reduce_workspace_bytes(n))
    ws = _ws(nbytes, x.device)
    px, ldx = _rows(x, "x")
    L.check(lib.jm_rows_colsum(M, _ptr(m_dev), n, px, ldx, _ptr(out), 0, _ptr(ws), nbytes, L.stream_ptr()), "rows_colsum")
    return out


def relu_mask_(dy: torch.Tensor, y: torch.Tensor, m_dev: Optional[torch.Tensor] = None) -> torch.Tensor:
    pdy, ldd = _rows(dy, "dy")
    py, ldy = _rows(y, "y")
    L.check(L.load().jm_rows_relu_mask(dy.shape[0], _ptr(m_dev), dy.shape[1], pdy, ldd, py, ldy, L.stream_ptr()), "rows_relu_mask")
    return dy


class _RowsMLP(Function):
    """a chain of dense layers on rows: x = [x1 | x2] -> act_0(W_0 x + b_0) -> ... ; acts[l] in {0 none, 1 ReLU, 2 tanh}.
    apply(x1, x2 or None, acts, W_0, b_0, W_1, b_1, ...) with b_l a tensor or None"""

    @staticmethod
    def forward(ctx, x1, x2, acts, *wb):
        nl = len(acts)
        assert len(wb) == 2 * nl
        ys = []
        cur, cur2 = x1, x2
        for l in range(nl):
            W, b = wb[2 * l], wb[2 * l + 1]
            y = linear_forward(cur, W, b, acts[l], cur2)
            ys.append(y)
            cur, cur2 = y, None
        ctx.acts = tuple(acts)
        ctx.has_x2 = x2 is not None
        ctx.has_b = tuple(b is not None for b in wb[1::2])
        ctx.save_for_backward(x1, *( [x2] if x2 is not None else []), *wb[0::2], *ys)
        return ys[-1]

    @staticmethod
    def backward(ctx, dout):
        acts, nl = ctx.acts, len(ctx.acts)
        saved = list(ctx.saved_tensors)
        x1 = saved.pop(0)
        x2 = saved.pop(0) if ctx.has_x2 else None
        Ws, ys = saved[:nl], saved[nl:]
        dy = dout.contiguous()
        if dy.data_ptr() == dout.data_ptr():
            dy = dy.clone()                      # masked in place below: never the caller's gradient buffer
        grads = [None] * (2 * nl)
        need_x1, need_x2 = ctx.needs_input_grad[0], ctx.has_x2 and ctx.needs_input_grad[1]
        # gradient w.r.t. the last layer's pre-activation
        if acts[-1] == 1:
            relu_mask_(dy, ys[-1])
        elif acts[-1] == 2:
            dy = dy * (1.0 - ys[-1] * ys[-1])
        dx1 = dx2 = None
        for l in range(nl - 1, -1, -1):
            ins = [ys[l - 1]] if l > 0 else ([x1, x2] if x2 is not None else [x1])
            dw, db = linear_wgrad(dy, ins, want_bias=ctx.has_b[l])
            grads[2 * l], grads[2 * l + 1] = dw, db
            if l > 0:
                a = acts[l - 1]
                nxt = linear_dgrad(dy, Ws[l], 0, ys[l - 1].shape[1], mask=ys[l - 1] if a == 1 else None)
                if a == 2:
                    nxt = nxt * (1.0 - ys[l - 1] * ys[l - 1])
                dy = nxt
            else:
                if need_x1:
                    dx1 = linear_dgrad(dy, Ws[0], 0, x1.shape[1])
                if need_x2:
                    dx2 = linear_dgrad(dy, Ws[0], x1.shape[1], x2.shape[1])
        return (dx1, dx2, None, *grads)


def rows_mlp(x1: torch.Tensor, layers: Sequence[Tuple[torch.Tensor, Optional[torch.Tensor]]], acts: Sequence[int],
             x2: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[x1 | x2] (M, k1 + k2) through the dense layers [(W (n, k), b (n) or None), ...] with activations acts"""
    flat = []
    for W, b in layers:
        flat += [W, b]
    return _RowsMLP.apply(x1, x2, tuple(int(a) for a in acts), *flat)


# ---------------------------------------------------------------------------------------------------- set abstraction on rows
class RowsPlan:
    """the distinct (centre, neighbour) rows of a batch of ball-query groups (csrc/rows_ops.hip: sa_rows_plan)"""

    def __init__(self, idx: torch.Tensor, n_per_set: int, canon: Optional[torch.Tensor] = None):
        """idx (S, M, ns) int32 neighbour lists local to their set of n_per_set points; canon (S, n_per_set) int32 or None"""
        S, M, ns = idx.shape
        idx = idx.contiguous()
        dev = idx.device
        G = S * M
        self.groups, self.max_rows, self.ns = G, G * ns, ns
        self.d = torch.empty((G,), dtype=_i32, device=dev)
        self.offsets = torch.empty((G + 1,), dtype=_i32, device=dev)
        self.row_point = torch.empty((G * ns,), dtype=_i32, device=dev)
        self.row_group = torch.empty((G * ns,), dtype=_i32, device=dev)
        L.check(L.load().jm_sa_rows_plan(G, ns, L.dev(idx, _i32, "idx"), L.dev(canon.contiguous(), _i32, "canon") if canon is not None else None,
                                         int(n_per_set), M, _ptr(self.d), _ptr(self.offsets), _ptr(self.row_point), _ptr(self.row_group),
                                         L.stream_ptr()), "sa_rows_plan")
        self.rows_dev = self.offsets[G:]          # (1,) int32 view: the row count, in device memory


class _SaScale(Function):
    """QueryAndGroup + SharedMLP + max-pool of one scale (pointnet2_modules.py:46-55) on the plan's rows.
    apply(f (P, C) or None, xyz (P, 3), ctr (G, 3) or None, plan, W1x (H1, 3), W1f (H1, C) or None, b1, W2, b2, ..., WL, bL)
    -> pooled (G, C_L).  Every layer is conv + ReLU (BatchNorm folded by the caller)."""

    @staticmethod
    def forward(ctx, f, xyz, ctr, plan, w1x, w1f, b1, *wb):
        lib = L.load()
        R, rdev = plan.max_rows, plan.rows_dev
        H1 = w1x.shape[0]
        dev = xyz.device
        u = None
        if f is not None:
            u = linear_forward(f, w1f, b1, 0)                             # per POINT: (P, H1)
        h = torch.empty((R, H1), dtype=_f32, device=dev)
        L.check(lib.jm_sa_rows_h1(R, _ptr(rdev), H1, _ptr(u), H1, _ptr(b1), L.dev(w1x.contiguous(), _f32, "w1x"), L.dev(xyz, _f32, "xyz"),
                                  L.dev(ctr, _f32, "ctr") if ctr is not None else None, _ptr(plan.row_point), _ptr(plan.row_group), _ptr(h), H1,
                                  L.stream_ptr()), "sa_rows_h1")
        hs = [h]
        for l in range(0, len(wb), 2):
            h = linear_forward(h, wb[l], wb[l + 1], 1, m_dev=rdev)
            hs.append(h)
        C = h.shape[1]
        out = torch.empty((plan.groups, C), dtype=_f32, device=dev)
        argrow = torch.empty((plan.groups, C), dtype=_i32, device=dev)
        L.check(lib.jm_sa_rows_pool(plan.groups, C, _ptr(h), C, _ptr(plan.offsets), _ptr(out), C, _ptr(argrow), L.stream_ptr()), "sa_rows_pool")
        ctx.plan = plan
        ctx.has_f, ctx.has_ctr = f is not None, ctr is not None
        ctx.nl = len(wb) // 2
        keep = [xyz, w1x, out, argrow] + ([f, w1f] if f is not None else []) + ([ctr] if ctr is not None else []) + list(wb[0::2]) + hs[:-1]
        ctx.save_for_backward(*keep)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = L.load()
        plan = ctx.plan
        R, rdev = plan.max_rows, plan.rows_dev
        saved = list(ctx.saved_tensors)
        xyz, w1x, out, argrow = saved[:4]
        saved = saved[4:]
        f = w1f = ctr = None
        if ctx.has_f:
            f, w1f = saved[:2]
            saved = saved[2:]
        if ctx.has_ctr:
            ctr = saved.pop(0)
        Ws, hs = saved[:ctx.nl], saved[ctx.nl:]           # hs[0] = H1 ... hs[nl - 1] = input of the last layer
        dev = xyz.device
        dout = dout.contiguous()
        C = out.shape[1]
        dh = torch.empty((R, C), dtype=_f32, device=dev)
        L.check(lib.jm_sa_rows_pool_grad(R, _ptr(rdev), C, _ptr(dout), C, _ptr(out), C, _ptr(argrow), _ptr(plan.row_group), _ptr(dh), C,
                                         L.stream_ptr()), "sa_rows_pool_grad")
        grads_wb = [None] * (2 * ctx.nl)
        for l in range(ctx.nl - 1, -1, -1):
            x_in = hs[l]
            dw, db = linear_wgrad(dh, [x_in], True, m_dev=rdev)
            grads_wb[2 * l], grads_wb[2 * l + 1] = dw, db
            dh = linear_dgrad(dh, Ws[l], 0, x_in.shape[1], mask=x_in, m_dev=rdev)
        # dh = gradient w.r.t. the first layer's pre-activation on the rows
        H1 = dh.shape[1]
        dw1x = torch.empty((H1, 3), dtype=_f32, device=dev)
        nbytes = int(lib.jm_rows_reduce_workspace_bytes(3 * H1))
        ws = _ws(nbytes, dev)
        L.check(lib.jm_sa_rows_xyz_wgrad(R, _ptr(rdev), H1, _ptr(dh), H1, _ptr(xyz), _ptr(ctr), _ptr(plan.row_point), _ptr(plan.row_group),
                                         _ptr(dw1x), 0, _ptr(ws), nbytes, L.stream_ptr()), "sa_rows_xyz_wgrad")
        db1 = colsum(dh, m_dev=rdev)
        df = dw1f = None
        if ctx.has_f:
            du = torch.zeros((f.shape[0], H1), dtype=_f32, device=dev)
            L.check(lib.jm_sa_rows_scatter_add(R, _ptr(rdev), H1, _ptr(dh), H1, _ptr(plan.row_point), _ptr(du), H1, L.stream_ptr()),
                    "sa_rows_scatter_add")
            dw1f, _ = linear_wgrad(du, [f], False)
            if ctx.needs_input_grad[0]:
                df = linear_dgrad(du, w1f, 0, f.shape[1])
        return (df, None, None, None, dw1x, dw1f, db1, *grads_wb)


def sa_scale_rows(f: Optional[torch.Tensor], xyz: torch.Tensor, ctr: Optional[torch.Tensor], plan: RowsPlan,
                  layers: Sequence[Tuple[torch.Tensor, torch.Tensor]]) -> torch.Tensor:
    """one set-abstraction scale on rows: f (P, C) point features (None: xyz only), xyz (P, 3) flat points, ctr (G, 3) group centres
    (None: GroupAll, coordinates not re-centred), layers = [(W (out, in), b)] with layer 0's input = [xyz (3) ; f (C)] as
    QueryAndGroup concatenates them (pointnet2_utils.py:259-269) -> (G, C_out)"""
    W1, b1 = layers[0]
    w1x = W1[:, :3].contiguous()
    w1f = W1[:, 3:].contiguous() if f is not None else None
    flat = []
    for W, b in layers[1:]:
        flat += [W, b]
    return _SaScale.apply(f, xyz, ctr, plan, w1x, w1f, b1, *flat)


# ---------------------------------------------------------------------------------------------------- interpolation / gather on rows
class _ThreeInterpolateRows(Function):
    @staticmethod
    def forward(ctx, known, idx, weight, B, n, m):
        C = known.shape[1]
        out = torch.empty((B * n, C), dtype=_f32, device=known.device)
        pk, ldk = _rows(known, "known")
        L.check(L.load().jm_three_interpolate_rows(B, n, m, C, pk, ldk, L.dev(idx, _i32, "idx"), L.dev(weight, _f32, "weight"), _ptr(out), C,
                                                   L.stream_ptr()), "three_interpolate_rows")
        ctx.save_for_backward(idx, weight)
        ctx.dims = (B, n, m, C)
        return out

    @staticmethod
    def backward(ctx, dout):
        idx, weight = ctx.saved_tensors
        B, n, m, C = ctx.dims
        dout = dout.contiguous()
        dk = torch.zeros((B * m, C), dtype=_f32, device=dout.device)
        L.check(L.load().jm_three_interpolate_rows_grad(B, n, m, C, _ptr(dout), C, _ptr(idx), _ptr(weight), _ptr(dk), C, L.stream_ptr()),
                "three_interpolate_rows_grad")
        return dk, None, None, None, None, None


def three_interpolate_rows(known: torch.Tensor, idx: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    """known (B m, C) rows, idx / weight (B, n, 3) -> (B n, C): pointnet2_utils.three_interpolate on rows"""
    B, n, _ = idx.shape
    return _ThreeInterpolateRows.apply(known, idx.contiguous(), weight.contiguous(), B, n, known.shape[0] // B)


class _FeatureGatherRows(Function):
    @staticmethod
    def forward(ctx, fmap, xy):
        B, C, H, W = fmap.shape
        if not fmap.is_contiguous(memory_format=torch.channels_last):
            fmap = fmap.contiguous(memory_format=torch.channels_last)
        xy = xy.contiguous()
        N = xy.shape[1]
        out = torch.empty((B * N, C), dtype=_f32, device=fmap.device)
        if not fmap.is_cuda:
            raise RuntimeError("feature_gather_rows: feature_map must be a GPU tensor (no CPU path)")
        L.check(L.load().jm_feature_gather_rows(B, C, H, W, N, _ptr(fmap), L.dev(xy, _f32, "xy"), _ptr(out), C, L.stream_ptr()),
                "feature_gather_rows")
        ctx.save_for_backward(xy)
        ctx.dims = (B, C, H, W, N)
        return out

    @staticmethod
    def backward(ctx, dout):
        (xy,) = ctx.saved_tensors
        B, C, H, W, N = ctx.dims
        dout = dout.contiguous()
        dmap = torch.empty((B, C, H, W), dtype=_f32, device=dout.device, memory_format=torch.channels_last).zero_()
        L.check(L.load().jm_feature_gather_rows_grad(B, C, H, W, N, _ptr(dout), C, _ptr(xy), _ptr(dmap), L.stream_ptr()),
                "feature_gather_rows_grad")
        return dmap, None


def feature_gather_rows(feature_map: torch.Tensor, xy: torch.Tensor) -> torch.Tensor:
    """feature_map (B, C, H, W) (channels-last memory), xy (B, N, 2) in [-1, 1] -> (B N, C): backbone.py:79-89 on rows"""
    return _FeatureGatherRows.apply(feature_map, xy)
