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


def linear_wgrad(dy: torch.Tensor, xs: Sequence[torch.Tensor], want_bias: bool = True, m_dev: Optional[torch.Tensor] = None,
                 out: Optional[torch.Tensor] = None, col0: int = 0):
    """(dw (N, sum k_i) = dy^T [x_1 | x_2 | ...], dbias (N) or None); out / col0: write the block into columns col0 .. of an
    existing (N, >= col0 + sum k_i) tensor instead"""
    lib = L.load()
    M, n = dy.shape
    ktot = sum(x.shape[1] for x in xs)
    dw = out if out is not None else torch.empty((n, ktot), dtype=_f32, device=dy.device)
    db = torch.empty((n,), dtype=_f32, device=dy.device) if want_bias else None
    pdy, lddy = _rows(dy, "dy")
    off = col0
    ktot = dw.shape[1]
    for i, x in enumerate(xs):
        k = x.shape[1]
        px, ldx = _rows(x, "x")
        nbytes = int(lib.jm_rows_wgrad_workspace_bytes(M, n, k))
        ws = _ws(nbytes, dy.device) if nbytes else None
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


def relu_mask(dy: torch.Tensor, y: torch.Tensor, m_dev: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y > 0 ? dy : 0 as a new tensor (dy = a gradient autograd handed in: never modified)"""
    out = torch.empty_like(y)
    pdy, ldd = _rows(dy, "dy")
    py, ldy = _rows(y, "y")
    L.check(L.load().jm_rows_relu_mask(dy.shape[0], _ptr(m_dev), dy.shape[1], pdy, ldd, py, ldy, _ptr(out), y.shape[1], L.stream_ptr()),
            "rows_relu_mask")
    return out


def _vp(t):
    return t.data_ptr() if t is not None else None


def _wgrad_ws_bytes(lib, M, pairs) -> int:
    return max([int(lib.jm_rows_wgrad_workspace_bytes(M, n, k)) for n, k in pairs] + [0])


class _RowsMLP(Function):
    """a chain of dense layers on rows: x = [x1 | x2] -> act_0(W_0 x + b_0) -> ... ; acts[l] in {0 none, 1 ReLU, 2 tanh}.
    apply(x1, x2 or None, acts, W_0, b_0, W_1, b_1, ...) with b_l a tensor or None.  ONE C call per direction
    (csrc/rows_chain.hip: jm_rows_mlp_forward / _backward)."""

    @staticmethod
    def _desc(x1, x2, acts, Ws, bs, ys):
        nl = len(acts)
        d = L.RowsMlp()
        d.nl, d.m, d.m_dev = nl, x1.shape[0], None
        d.k1, d.k2 = x1.shape[1], (x2.shape[1] if x2 is not None else 0)
        p1, d.ldx1 = _rows(x1, "x1")
        d.x1 = p1.value
        if x2 is not None:
            p2, d.ldx2 = _rows(x2, "x2")
            d.x2 = p2.value
        kin = d.k1 + d.k2
        for l in range(nl):
            W = Ws[l]
            if W.shape[1] != kin:
                raise ValueError(f"layer {l}: weight of {W.shape[1]} input channels for an input of {kin}")
            pw, ldw = _rows(W, f"W{l}")
            d.widths[l], d.acts[l], d.w[l], d.ldw[l] = W.shape[0], int(acts[l]), pw.value, ldw
            d.b[l] = _vp(bs[l])
            d.y[l] = ys[l].data_ptr()
            kin = W.shape[0]
        return d

    @staticmethod
    def forward(ctx, x1, x2, acts, *wb):
        nl = len(acts)
        assert len(wb) == 2 * nl and nl <= L.ROWS_MAX_LAYERS
        Ws, bs = wb[0::2], wb[1::2]
        M = x1.shape[0]
        ys = [torch.empty((M, W.shape[0]), dtype=_f32, device=x1.device) for W in Ws]
        d = _RowsMLP._desc(x1, x2, acts, Ws, bs, ys)
        L.check(L.load().jm_rows_mlp_forward(ctypes.byref(d), L.stream_ptr()), "rows_mlp_forward")
        ctx.acts = tuple(acts)
        ctx.has_x2 = x2 is not None
        ctx.has_b = tuple(b is not None for b in bs)
        ctx.save_for_backward(x1, *([x2] if x2 is not None else []), *Ws, *ys)
        return ys[-1]

    @staticmethod
    def backward(ctx, dout):
        lib = L.load()
        acts, nl = ctx.acts, len(ctx.acts)
        saved = list(ctx.saved_tensors)
        x1 = saved.pop(0)
        x2 = saved.pop(0) if ctx.has_x2 else None
        Ws, ys = saved[:nl], saved[nl:]
        dev = x1.device
        M = x1.shape[0]
        if not (dout.stride(1) == 1 and dout.stride(0) % 4 == 0 and dout.data_ptr() % 16 == 0):
            dout = dout.contiguous()
        d = _RowsMLP._desc(x1, x2, acts, Ws, [None] * nl, ys)
        g = L.RowsMlpGrad()
        g.dout, g.lddout = dout.data_ptr(), dout.stride(0) if M > 1 else dout.shape[1]
        dws = [torch.empty_like(W) if W.is_contiguous() else torch.empty(W.shape, dtype=_f32, device=dev) for W in Ws]
        dbs = [torch.empty((W.shape[0],), dtype=_f32, device=dev) if hb else None for W, hb in zip(Ws, ctx.has_b)]
        for l in range(nl):
            g.dw[l], g.lddw[l], g.db[l] = dws[l].data_ptr(), dws[l].shape[1], _vp(dbs[l])
        need_x1, need_x2 = ctx.needs_input_grad[0], ctx.has_x2 and ctx.needs_input_grad[1]
        dx1 = torch.empty_like(x1) if need_x1 else None
        dx2 = torch.empty_like(x2) if need_x2 else None
        g.dx1, g.dx2 = _vp(dx1), _vp(dx2)
        wmax = max(W.shape[0] for W in Ws)
        pairs = [(Ws[l].shape[0], Ws[l].shape[1]) for l in range(1, nl)] + [(Ws[0].shape[0], x1.shape[1])] + \
                ([(Ws[0].shape[0], x2.shape[1])] if x2 is not None else [])
        nbytes = _wgrad_ws_bytes(lib, M, pairs)
        grads = []
        for l in range(nl):
            grads += [dws[l], dbs[l]]
        scratch = torch.empty((2, M, wmax), dtype=_f32, device=dev)
        g.scratch[0], g.scratch[1] = scratch[0].data_ptr(), scratch[1].data_ptr()
        ws = _ws(nbytes, dev) if nbytes else None
        g.ws, g.ws_bytes = _vp(ws), nbytes
        L.check(lib.jm_rows_mlp_backward(ctypes.byref(d), ctypes.byref(g), L.stream_ptr()), "rows_mlp_backward")
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

    @classmethod
    def from_tensors(cls, d: torch.Tensor, offsets: torch.Tensor, row_point: torch.Tensor, row_group: torch.Tensor, groups: int,
                     ns: int) -> "RowsPlan":
        """a plan whose four device tensors were made elsewhere (e.g. by a replayed graph: tools/quarantine/train_graphs.py)"""
        p = cls.__new__(cls)
        p.groups, p.max_rows, p.ns = int(groups), int(groups) * int(ns), int(ns)
        p.d, p.offsets, p.row_point, p.row_group = d, offsets, row_point, row_group
        p.rows_dev = offsets[groups:]
        return p


class _SaLevel(Function):
    """QueryAndGroup + SharedMLP + max-pool (pointnet2_modules.py:46-55) of ALL scales of one set-abstraction level on their plans'
    rows; the scales' pooled features land in their channel slices of ONE (G, sum C_k) tensor (the torch.cat of :54 without a copy).
    apply(f (P, C) or None, xyz (P, 3), ctr (G, 3) or None, plans, nlayers, W1_0, b1_0, W2_0, b2_0, ..., W1_1, b1_1, ...) with
    W1_k (H1, 3 + C) = the first layer's weight on [xyz ; f] as QueryAndGroup concatenates them.  Every layer is conv + ReLU.
    One C call per scale and direction (csrc/rows_chain.hip: jm_sa_scale_forward / _backward)."""

    @staticmethod
    def _desc(plan, nl, f, xyz, ctr, w1x, w1f, b1, Ws, bs, slab, widths, out, c0, argrow):
        d = L.SaScale()
        d.nl, d.groups, d.max_rows = nl, plan.groups, plan.max_rows
        d.rows_dev, d.offsets, d.row_point, d.row_group = plan.rows_dev.data_ptr(), plan.offsets.data_ptr(), plan.row_point.data_ptr(), plan.row_group.data_ptr()
        d.points, d.c = (f.shape[0], f.shape[1]) if f is not None else (xyz.shape[0], 0)
        if f is not None:
            pf, d.ldf = _rows(f, "f")
            d.f = pf.value
        d.xyz, d.ctr = xyz.data_ptr(), _vp(ctr)
        d.w1x, d.w1f, d.b1 = w1x.data_ptr(), _vp(w1f), b1.data_ptr()
        for l in range(nl):
            d.widths[l] = widths[l]
        for l in range(1, nl):
            d.w[l], d.b[l] = Ws[l - 1].data_ptr(), _vp(bs[l - 1]) if bs is not None else None
        # slab layout (floats): u (points, H1) | delta (R, 4) | h[0] (R, H1) | ... | h[nl - 1]
        base, R, H1 = slab.data_ptr(), plan.max_rows, widths[0]
        off = 0
        d.u = base
        off += d.points * H1 if f is not None else 0
        d.delta = base + 4 * off
        off += R * 4
        for l in range(nl):
            d.h[l] = base + 4 * off
            off += R * widths[l]
        d.out, d.ldo, d.argrow = out.data_ptr() + 4 * c0, out.shape[1], argrow.data_ptr()
        return d

    @staticmethod
    def _slab_elems(plan, nl, f, xyz, widths):
        return (f.shape[0] * widths[0] if f is not None else 0) + plan.max_rows * (4 + sum(widths[:nl]))

    @staticmethod
    def forward(ctx, f, xyz, ctr, plans, nlayers, *wb):
        lib = L.load()
        dev = xyz.device
        K, per = len(plans), 2 * nlayers
        G = plans[0].groups
        couts = [wb[k * per + per - 2].shape[0] for k in range(K)]
        out = torch.empty((G, sum(couts)), dtype=_f32, device=dev)
        L.dev(xyz, _f32, "xyz")
        if ctr is not None:
            L.dev(ctr, _f32, "ctr")
        keep, c0 = [], 0
        for k, plan in enumerate(plans):
            W1, b1 = wb[k * per], wb[k * per + 1]
            Ws = [wb[k * per + 2 * l].contiguous() for l in range(1, nlayers)]
            bs = [wb[k * per + 2 * l + 1] for l in range(1, nlayers)]
            widths = [W1.shape[0]] + [W.shape[0] for W in Ws]
            w1x = W1[:, :3].contiguous()
            w1f = W1[:, 3:].contiguous() if f is not None else None        # (aligned copy: the row kernels read 16-byte vectors)
            slab = torch.empty((_SaLevel._slab_elems(plan, nlayers, f, xyz, widths),), dtype=_f32, device=dev)
            argrow = torch.empty((G, couts[k]), dtype=_i32, device=dev)
            d = _SaLevel._desc(plan, nlayers, f, xyz, ctr, w1x, w1f, b1, Ws, bs, slab, widths, out, c0, argrow)
            L.check(lib.jm_sa_scale_forward(ctypes.byref(d), L.stream_ptr()), "sa_scale_forward")
            keep.append([slab, argrow, w1x] + ([w1f] if f is not None else []) + Ws)
            c0 += couts[k]
        ctx.plans, ctx.nlayers, ctx.couts, ctx.has_f, ctx.has_ctr = plans, nlayers, couts, f is not None, ctr is not None
        ctx.counts = [len(kk) for kk in keep]
        ctx.w1_cols = wb[0].shape[1]
        ctx.save_for_backward(out, xyz, *([ctr] if ctr is not None else []), *([f] if f is not None else []), *[t for kk in keep for t in kk])
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = L.load()
        saved = list(ctx.saved_tensors)
        out, xyz = saved.pop(0), saved.pop(0)
        ctr = saved.pop(0) if ctx.has_ctr else None
        f = saved.pop(0) if ctx.has_f else None
        dev = out.device
        nl, per = ctx.nlayers, 2 * ctx.nlayers
        dout = dout.contiguous()
        grads, c0, later = [], 0, []
        df = torch.empty_like(f) if (ctx.has_f and ctx.needs_input_grad[0]) else None
        for k, plan in enumerate(ctx.plans):
            mine, saved = saved[:ctx.counts[k]], saved[ctx.counts[k]:]
            slab, argrow, w1x = mine[0], mine[1], mine[2]
            mine = mine[3:]
            w1f = mine.pop(0) if ctx.has_f else None
            Ws = mine
            widths = [w1x.shape[0]] + [W.shape[0] for W in Ws]
            H1, R = widths[0], plan.max_rows
            b1_dummy = w1x          # (the backward never reads b1; the field must not be NULL)
            d = _SaLevel._desc(plan, nl, f, xyz, ctr, w1x, w1f, b1_dummy, Ws, None, slab, widths, out, c0, argrow)
            g = L.SaScaleGrad()
            g.dout, g.lddout = dout.data_ptr() + 4 * c0, dout.shape[1]
            dW1 = torch.empty((H1, ctx.w1_cols), dtype=_f32, device=dev)
            db1 = torch.empty((H1,), dtype=_f32, device=dev)
            dws = [torch.empty_like(W) for W in Ws]
            dbs = [torch.empty((W.shape[0],), dtype=_f32, device=dev) for W in Ws]
            wmax = max(widths)
            nsc = 2 * R * wmax + H1 * 4 + (f.shape[0] * H1 if f is not None else 0)
            scratch = torch.empty((nsc,), dtype=_f32, device=dev)
            base = scratch.data_ptr()
            g.scratch[0], g.scratch[1] = base, base + 4 * R * wmax
            g.dw4 = base + 8 * R * wmax
            g.du = base + 8 * R * wmax + 16 * H1 if f is not None else None
            g.dw1, g.db1 = dW1.data_ptr(), db1.data_ptr()
            for l in range(1, nl):
                g.dw[l], g.db[l] = dws[l - 1].data_ptr(), dbs[l - 1].data_ptr()
            g.df, g.df_accumulate = _vp(df), 1 if k > 0 else 0
            pairs = [(widths[l], widths[l - 1]) for l in range(1, nl)] + [(H1, 4)]
            nbytes = _wgrad_ws_bytes(lib, R, pairs)
            if f is not None:
                nbytes = max(nbytes, int(lib.jm_rows_wgrad_workspace_bytes(f.shape[0], H1, f.shape[1])))
            gk = [dW1, db1]
            for l in range(1, nl):
                gk += [dws[l - 1], dbs[l - 1]]
            grads += gk
            c0 += ctx.couts[k]
            ws = _ws(nbytes, dev) if nbytes else None
            g.ws, g.ws_bytes = _vp(ws), nbytes
            L.check(lib.jm_sa_scale_backward(ctypes.byref(d), ctypes.byref(g), L.stream_ptr()), "sa_scale_backward")
        return (df, None, None, None, None, *grads)


def sa_level_rows(f: Optional[torch.Tensor], xyz: torch.Tensor, ctr: Optional[torch.Tensor], plans: Sequence[RowsPlan],
                  scales: Sequence[Sequence[Tuple[torch.Tensor, torch.Tensor]]]) -> torch.Tensor:
    """all scales of one set-abstraction level on rows: f (P, C) point features (None: xyz only), xyz (P, 3) flat points, ctr (G, 3)
    group centres (None: GroupAll, coordinates not re-centred), scales[k] = [(W (out, in), b)] of scale k with layer 0's input =
    [xyz (3) ; f (C)] as QueryAndGroup concatenates them (pointnet2_utils.py:259-269) -> (G, sum_k C_out_k)"""
    nl = len(scales[0])
    assert all(len(sc) == nl for sc in scales) and nl >= 2
    flat = []
    for sc in scales:
        for W, b in sc:
            flat += [W, b]
    return _SaLevel.apply(f, xyz, ctr, tuple(plans), nl, *flat)


def sa_scale_rows(f, xyz, ctr, plan: RowsPlan, layers) -> torch.Tensor:
    """one scale (see sa_level_rows)"""
    return sa_level_rows(f, xyz, ctr, [plan], [layers])


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


# ---------------------------------------------------------------------------------------------------- BatchNorm folding
# A convolution library may read PAST the end of a small weight tensor: MIOpen's data-gradient kernel for the 1 x 1 convolution of
# 16 -> 8 channels (2 x 96 x 320, fp32: the fusion convolution of DetectorConfig.tiny()) loads whole tiles of output channels from its
# 512-byte weight without a bound — a device memory fault whenever that tensor is the last block of a caching-allocator segment and
# the next page is unmapped (tools/miopen_oob_probe.py reproduces it with plain torch; DESIGN.md section 6: the "rows backward fault"
# of round 5 was this, not stream ordering).  Folded weights are temporaries made every step, so they would land anywhere: they are
# carved from ONE slab with slack behind the last of them — an over-read of a folded copy stays inside mapped, owned memory (and the
# step makes one allocation instead of one per convolution).
FOLD_SLAB_SLACK = 64 << 10       # bytes behind the last folded weight (the over-read seen: < 4 x the tensor; the largest folded weight
                                 # of the reference network is 2.4 MB and sits in front of others)


def _slab_like(ws):
    """[empty tensor laid out like w for w in ws] as views of one float32 slab: 256-byte aligned, FOLD_SLAB_SLACK bytes behind the last"""
    offs, total = [], 0
    for w in ws:
        offs.append(total)
        total += (w.numel() + 63) // 64 * 64
    slab = torch.empty((total + FOLD_SLAB_SLACK // 4,), dtype=_f32, device=ws[0].device)
    return [slab[o:o + w.numel()].as_strided(w.shape, w.stride()) for o, w in zip(offs, ws)]


class _FoldAll(Function):
    """wf_l = w_l * s[soff_l + row] for every (convolution, BatchNorm) pair of a network in ONE launch, and its backward in one more
    (csrc/rows_ops.hip: fold_bn_multi).  apply(s (sum C,), soffs, w_0, w_1, ...) -> (wf_0, wf_1, ...), each wf_l laid out like w_l."""

    @staticmethod
    def _table(ts):
        n = len(ts)
        return (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])

    @staticmethod
    def forward(ctx, s, soffs, *ws):
        n = len(ws)
        for w in ws:
            if not (w.is_contiguous() or (w.dim() == 4 and w.is_contiguous(memory_format=torch.channels_last))):
                raise RuntimeError("fold: weights must be dense (contiguous or channels-last)")
        outs = _slab_like(ws)
        rows = (ctypes.c_int * n)(*[w.shape[0] for w in ws])
        cols = (ctypes.c_int * n)(*[w.numel() // w.shape[0] for w in ws])
        so = (ctypes.c_int * n)(*soffs)
        L.check(L.load().jm_fold_bn_multi(n, _FoldAll._table(ws), _FoldAll._table(outs), rows, cols, so, L.dev(s, _f32, "s"), L.stream_ptr()),
                "fold_bn_multi")
        ctx.save_for_backward(s, *ws)
        ctx.soffs = soffs
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dwfs):
        s, *ws = ctx.saved_tensors
        live = [i for i, g in enumerate(dwfs) if g is not None]
        ds = torch.zeros_like(s)
        dws = [None] * len(ws)
        if live:
            gs = []
            for i in live:
                g, w = dwfs[i], ws[i]
                if g.shape != w.shape:
                    g = g.reshape(w.shape)
                if g.stride() != w.stride():         # same memory order as the weight (channels-last 3x3 kernels)
                    g = torch.empty_like(w).copy_(g)
                gs.append(g)
                dws[i] = torch.empty_like(w)
            n = len(live)
            rows = (ctypes.c_int * n)(*[ws[i].shape[0] for i in live])
            cols = (ctypes.c_int * n)(*[ws[i].numel() // ws[i].shape[0] for i in live])
            so = (ctypes.c_int * n)(*[ctx.soffs[i] for i in live])
            L.check(L.load().jm_fold_bn_multi_grad(n, _FoldAll._table(gs), _FoldAll._table([ws[i] for i in live]),
                                                   _FoldAll._table([dws[i] for i in live]), rows, cols, so, _ptr(s), _ptr(ds), L.stream_ptr()),
                    "fold_bn_multi_grad")
        return (ds, None, *dws)


def fold_all(s: torch.Tensor, soffs: Sequence[int], weights: Sequence[torch.Tensor]):
    return _FoldAll.apply(s, tuple(int(o) for o in soffs), *weights)
