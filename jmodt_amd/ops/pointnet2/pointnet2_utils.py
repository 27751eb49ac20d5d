"""PointNet++ operator API on the gfx950 kernels.

Mirror of jmodt/ops/pointnet2/pointnet2_utils.py: the same callables with the same argument
meaning and return values —
    farthest_point_sample(xyz, npoint)            (pointnet2_utils.py:10-36)
    gather_operation(features, idx)               (:39-73)
    three_nn(unknown, known) -> (dist, idx)       (:76-105)
    three_interpolate(features, idx, weight)      (:108-153)
    grouping_operation(features, idx)             (:156-197)
    ball_query(radius, nsample, xyz, new_xyz)     (:200-228)
    QueryAndGroup / GroupAll                      (:231-290)
— all `torch.autograd.Function.apply`, backward defined for gather / group / interpolate only.
Differences by design: outputs are allocated on the INPUT's device with torch.empty (the
reference hard-codes `torch.cuda.*Tensor` on the current device), launches go to torch's current
stream, and errors surface as Python exceptions instead of exit().
"""
from typing import Optional, Tuple

import ctypes

import torch
import torch.nn as nn
from torch.autograd import Function

from ... import _lib as L
from ...profile import prof
from ...ext import pointnet2_cuda

_f32, _i32 = torch.float32, torch.int32


def _need(t: torch.Tensor, what: str):
    assert t.is_contiguous(), f"{what} must be contiguous"


class _FurthestPointSampling(Function):
    @staticmethod
    def forward(ctx, xyz: torch.Tensor, npoint: int) -> torch.Tensor:
        """xyz (B, N, 3) -> (B, npoint) int32 indices of the iteratively farthest points"""
        _need(xyz, "xyz")
        B, N, _ = xyz.size()
        idx = torch.empty((B, npoint), dtype=_i32, device=xyz.device)
        temp = torch.full((B, N), 1e10, dtype=_f32, device=xyz.device)
        # (the shim picks the co-operative multi-workgroup kernel for clouds of more than 16384 points)
        pointnet2_cuda.farthest_point_sampling_wrapper(B, N, npoint, xyz, temp, idx)
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, grad=None):
        return None, None


farthest_point_sample = _FurthestPointSampling.apply
furthest_point_sample = farthest_point_sample


@torch.no_grad()
def farthest_point_sample_xyz(xyz: torch.Tensor, npoint: int):
    """xyz (B, N, 3) -> (idx (B, npoint) int32, new_xyz (B, npoint, 3)) in ONE call: the sampling and the
    coordinate gather that pointnet2_modules.py:35-39 spells as transpose + gather_operation + transpose.
    Coordinates carry no gradient in the reference either (only features do), so this is a drop-in for it."""
    _need(xyz, "xyz")
    B, N, _ = xyz.size()
    idx = torch.empty((B, npoint), dtype=_i32, device=xyz.device)
    new_xyz = torch.empty((B, npoint, 3), dtype=_f32, device=xyz.device)
    # no temp buffer: the kernels start from the 1e10 fill themselves (one fill launch and 4 B / point less on
    # the sampling chain); only clouds beyond the co-operative kernel's limit still need it
    temp = torch.full((B, N), 1e10, dtype=_f32, device=xyz.device) if N > 131072 else None
    pointnet2_cuda.farthest_point_sampling_wrapper(B, N, npoint, xyz, temp, idx, new_xyz)
    if 16384 < N <= 131072 and npoint > 0:
        # the co-operative kernel (several workgroups per cloud exchanging records) gives a cloud up when a peer never shows
        # (csrc/fps.hip: the row becomes -1, its centres NaN) instead of hanging the GPU.  Nothing downstream may index with -1:
        # the row is clamped to index 0 HERE and the failure is kept as a device flag that `check_fps_failures()` turns into an
        # exception at the caller's next synchronisation point — no host sync on the sampling chain
        bad = idx[:, :1] < 0
        flag = bad.any()
        ev = torch.cuda.Event()
        ev.record()                               # on the stream that produced the flag (an FPS side stream under the prefetch)
        FPS_FAILED.append((flag, ev))
        if len(FPS_FAILED) > FPS_FAILED_KEEP:     # a caller that never checks: the oldest flags are dropped, loudly
            import warnings
            warnings.warn(f"furthest_point_sample: {len(FPS_FAILED) - FPS_FAILED_KEEP} unchecked co-operative-kernel failure "
                          f"flag(s) dropped; call check_fps_failures() at a synchronisation point", RuntimeWarning, stacklevel=2)
            del FPS_FAILED[:-FPS_FAILED_KEEP]
        idx = torch.where(bad, torch.zeros_like(idx), idx)
        new_xyz = torch.where(bad.unsqueeze(-1), xyz[:, :1].expand(-1, npoint, -1), new_xyz)
    return idx, new_xyz


FPS_FAILED = []      # (device bool, event recorded behind it on ITS stream) of the co-operative FPS launches not yet checked
FPS_FAILED_KEEP = 64


def check_fps_failures(wait: bool = False) -> None:
    """raise if a co-operative furthest_point_sample launch since the last call gave a cloud up (exchange time-out: a broken
    device partition, or a launch overlapped with work that kept its peer workgroups off the machine for seconds).
    The flags are written on the stream that ran the sampling (a side stream under the next-batch prefetch), not on the caller's:
    only flags whose own event has COMPLETED are read (and consumed); the others stay listed for the next call — wait=True
    synchronises on them instead.  Call it where the host waits for the device anyway (DetectionCache.counts_host does)."""
    ready, pending = [], []
    for flag, ev in FPS_FAILED:
        if wait:
            ev.synchronize()
        (ready if wait or ev.query() else pending).append((flag, ev))
    FPS_FAILED[:] = pending
    flags = [f for f, _ in ready]
    if flags and bool(torch.stack(flags).any().item()):
        raise RuntimeError("furthest_point_sample: the co-operative kernel timed out waiting for a peer workgroup; the sampled "
                           "indices of at least one cloud are invalid (they were clamped to 0)")


class _GatherOperation(Function):
    @staticmethod
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        """features (B, C, N), idx (B, npoint) -> (B, C, npoint)"""
        _need(features, "features"); _need(idx, "idx")
        B, npoint = idx.size()
        _, C, N = features.size()
        out = torch.empty((B, C, npoint), dtype=_f32, device=features.device)
        L.check(L.load().jm_gather_points(B, C, N, npoint, L.dev(features, _f32, "features"), L.dev(idx, _i32, "idx"),
                                          L.dev(out, _f32, "out"), L.stream_ptr()), "gather_operation")
        ctx.save_for_backward(idx)
        ctx.dims = (C, N)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        C, N = ctx.dims
        B, npoint = idx.size()
        grad_features = torch.zeros((B, C, N), dtype=_f32, device=grad_out.device)
        g = grad_out.contiguous()
        L.check(L.load().jm_gather_points_grad(B, C, N, npoint, L.dev(g, _f32, "grad_out"), L.dev(idx, _i32, "idx"),
                                               L.dev(grad_features, _f32, "grad_features"), L.stream_ptr()),
                "gather_operation.backward")
        return grad_features, None


gather_operation = _GatherOperation.apply


class _ThreeNN(Function):
    @staticmethod
    def forward(ctx, unknown: torch.Tensor, known: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """unknown (B, N, 3), known (B, M, 3) -> dist (B, N, 3) L2 distances, idx (B, N, 3)"""
        _need(unknown, "unknown"); _need(known, "known")
        B, N, _ = unknown.size()
        m = known.size(1)
        dist2 = torch.empty((B, N, 3), dtype=_f32, device=unknown.device)
        idx = torch.empty((B, N, 3), dtype=_i32, device=unknown.device)
        lib = L.load()
        ws_bytes = lib.jm_three_nn_workspace_bytes(B, N, m)       # hash grid over the known points where that pays (same output)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=unknown.device) if ws_bytes else None
        L.check(lib.jm_three_nn_ws(B, N, m, L.dev(unknown, _f32, "unknown"), L.dev(known, _f32, "known"),
                                   L.dev(dist2, _f32, "dist2"), L.dev(idx, _i32, "idx"),
                                   ctypes.c_void_p(ws.data_ptr()) if ws is not None else None, ws_bytes, L.stream_ptr()), "three_nn")
        ctx.mark_non_differentiable(idx)
        return torch.sqrt(dist2), idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None


three_nn = _ThreeNN.apply


@torch.no_grad()
def three_nn_weights(unknown: torch.Tensor, known: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """(idx (B, N, 3) int32, weight (B, N, 3)) = three_nn + the normalised inverse distances of PointnetFPModule.forward
    (pointnet2_modules.py:147-150: dist_recip = 1 / (dist + 1e-8), weight = dist_recip / sum) — search + ONE launch instead of
    search + sqrt + add + reciprocal + sum + div"""
    _need(unknown, "unknown"); _need(known, "known")
    B, N, _ = unknown.size()
    m = known.size(1)
    dist2 = torch.empty((B, N, 3), dtype=_f32, device=unknown.device)
    idx = torch.empty((B, N, 3), dtype=_i32, device=unknown.device)
    lib = L.load()
    ws_bytes = lib.jm_three_nn_workspace_bytes(B, N, m)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=unknown.device) if ws_bytes else None
    L.check(lib.jm_three_nn_ws(B, N, m, L.dev(unknown, _f32, "unknown"), L.dev(known, _f32, "known"), L.dev(dist2, _f32, "dist2"),
                               L.dev(idx, _i32, "idx"), ctypes.c_void_p(ws.data_ptr()) if ws is not None else None, ws_bytes, L.stream_ptr()),
            "three_nn")
    weight = torch.empty((B, N, 3), dtype=_f32, device=unknown.device)
    L.check(lib.jm_three_nn_weights(B * N, L.dev(dist2, _f32, "dist2"), L.dev(weight, _f32, "weight"), L.stream_ptr()), "three_nn_weights")
    return idx, weight


@torch.no_grad()
def gather_point_rows(src: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """src (B, n, w) float32, idx (B, m) int32 -> (B, m, w): src[b, idx[b, j], :] in one launch (backbone.py:170-171's
    torch.gather on the LI-Fusion pixel coordinates, without the int64 copy of the indices)"""
    src = src.contiguous()
    B, n, w = src.shape
    m = idx.shape[1]
    out = torch.empty((B, m, w), dtype=_f32, device=src.device)
    L.check(L.load().jm_gather_point_rows(B, n, m, w, L.dev(src, _f32, "src"), L.dev(idx.contiguous(), _i32, "idx"), L.dev(out, _f32, "out"),
                                          L.stream_ptr()), "gather_point_rows")
    return out


class _ThreeInterpolate(Function):
    @staticmethod
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
        """features (B, C, M), idx / weight (B, n, 3) -> (B, C, n)"""
        _need(features, "features"); _need(idx, "idx"); _need(weight, "weight")
        B, c, m = features.size()
        n = idx.size(1)
        feats = features.float()  # custom ops stay fp32 under autocast (pointnet2_utils.py:130)
        out = torch.empty((B, c, n), dtype=_f32, device=features.device)
        L.check(L.load().jm_three_interpolate(B, c, m, n, L.dev(feats, _f32, "features"), L.dev(idx, _i32, "idx"),
                                              L.dev(weight, _f32, "weight"), L.dev(out, _f32, "out"), L.stream_ptr()),
                "three_interpolate")
        ctx.save_for_backward(idx, weight)
        ctx.m = m
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight = ctx.saved_tensors
        B, c, n = grad_out.size()
        grad_features = torch.zeros((B, c, ctx.m), dtype=_f32, device=grad_out.device)
        g = grad_out.contiguous()
        L.check(L.load().jm_three_interpolate_grad(B, c, n, ctx.m, L.dev(g, _f32, "grad_out"), L.dev(idx, _i32, "idx"),
                                                   L.dev(weight, _f32, "weight"),
                                                   L.dev(grad_features, _f32, "grad_features"), L.stream_ptr()),
                "three_interpolate.backward")
        return grad_features, None, None


three_interpolate = _ThreeInterpolate.apply


class _GroupingOperation(Function):
    @staticmethod
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        """features (B, C, N), idx (B, npoint, nsample) -> (B, C, npoint, nsample)"""
        _need(features, "features"); _need(idx, "idx")
        B, npoint, nsample = idx.size()
        _, C, N = features.size()
        feats = features.float()
        out = torch.empty((B, C, npoint, nsample), dtype=_f32, device=features.device)
        L.check(L.load().jm_group_points(B, C, N, npoint, nsample, L.dev(feats, _f32, "features"),
                                         L.dev(idx, _i32, "idx"), L.dev(out, _f32, "out"), L.stream_ptr()),
                "grouping_operation")
        ctx.save_for_backward(idx)
        ctx.N = N
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        B, C, npoint, nsample = grad_out.size()
        grad_features = torch.zeros((B, C, ctx.N), dtype=_f32, device=grad_out.device)
        g = grad_out.contiguous()
        L.check(L.load().jm_group_points_grad(B, C, ctx.N, npoint, nsample, L.dev(g, _f32, "grad_out"),
                                              L.dev(idx, _i32, "idx"), L.dev(grad_features, _f32, "grad_features"),
                                              L.stream_ptr()), "grouping_operation.backward")
        return grad_features, None


grouping_operation = _GroupingOperation.apply


def _ball_query_workspace(B: int, N: int, device):
    """scratch of the hash-grid search (csrc/ball_query_grid.hip): bucket table + bucket-sorted points; (None, 0) where the
    library scans all points anyway (small clouds)"""
    nbytes = L.load().jm_ball_query_workspace_bytes(B, N)
    if nbytes == 0:
        return None, 0
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=device)
    if prof.enabled and prof.only is None:      # the launch counts its distance evaluations at the end of the workspace: kept for bench.py to read
        BQ_EVALS.append((prof._key("ball_query"), ws, L.load().jm_ball_query_evals_offset(B, N), B * N))
    return ws, nbytes


class BallQueryGrid:
    """the hash grid of a cloud, built ahead of the searches (jm_ball_query_grid_build): the build needs the points and the
    largest search radius only — e.g. on the FPS side stream, while the centres are still being sampled (pyramid.py).  None of
    it is needed for correctness: `ball_query(..., grid=None)` builds its own."""

    def __init__(self, xyz: torch.Tensor, cell_radius: float):
        _need(xyz, "xyz")
        self.B, self.N = xyz.shape[0], xyz.shape[1]
        self.cell_radius = float(cell_radius)
        lib = L.load()
        self.nbytes = lib.jm_ball_query_workspace_bytes(self.B, self.N)
        self.ws = None
        if self.nbytes and self.cell_radius > 0:
            self.ws = torch.empty((self.nbytes,), dtype=torch.uint8, device=xyz.device)
            L.check(lib.jm_ball_query_grid_build(self.B, self.N, self.cell_radius, L.dev(xyz, _f32, "xyz"),
                                                 ctypes.c_void_p(self.ws.data_ptr()), self.nbytes, L.stream_ptr()), "ball_query_grid_build")

    def matches(self, xyz: torch.Tensor) -> bool:
        return self.ws is not None and xyz.shape[0] == self.B and xyz.shape[1] == self.N

    def query(self, new_xyz, r0, ns0, r1=0.0, ns1=0):
        B, npoint = new_xyz.shape[0], new_xyz.shape[1]
        idx0 = torch.empty((B, npoint, ns0), dtype=_i32, device=new_xyz.device)      # (the kernels write 0 into the slots of an empty ball)
        idx1 = torch.empty((B, npoint, ns1), dtype=_i32, device=new_xyz.device) if ns1 else None
        if prof.enabled and prof.only is None:
            BQ_EVALS.append((prof._key("ball_query"), self.ws, L.load().jm_ball_query_evals_offset(self.B, self.N), self.B * self.N))
        L.check(L.load().jm_ball_query_grid_query(B, self.N, npoint, self.cell_radius, float(r0), ns0, float(r1), ns1,
                                                  L.dev(new_xyz, _f32, "new_xyz"), L.dev(idx0, _i32, "idx0"),
                                                  L.dev(idx1, _i32, "idx1") if idx1 is not None else None,
                                                  ctypes.c_void_p(self.ws.data_ptr()), self.nbytes, L.stream_ptr()), "ball_query_grid_query")
        return idx0, idx1


BQ_EVALS = []     # (profile scope, workspace, counter offset, B * N) of the grid searches launched while the profiler was on


def ball_query_evals():
    """{profile scope: (distance evaluations, launches)} of the recorded grid searches; clears the record (synchronises)"""
    torch.cuda.synchronize()
    out = {}
    for key, ws, off, _ in BQ_EVALS:
        n = int(ws[off:off + 256].view(torch.int64).sum().item())
        e, c = out.get(key, (0, 0))
        out[key] = (e + n, c + 1)
    BQ_EVALS.clear()
    return out


class _BallQuery(Function):
    @staticmethod
    def forward(ctx, radius: float, nsample: int, xyz: torch.Tensor, new_xyz: torch.Tensor) -> torch.Tensor:
        """xyz (B, N, 3), new_xyz (B, npoint, 3) -> (B, npoint, nsample) int32 neighbour indices"""
        _need(new_xyz, "new_xyz"); _need(xyz, "xyz")
        B, N, _ = xyz.size()
        npoint = new_xyz.size(1)
        idx = torch.empty((B, npoint, nsample), dtype=_i32, device=xyz.device)              # (empty balls come back as 0 from the kernel)
        ws, ws_bytes = _ball_query_workspace(B, N, xyz.device)
        L.check(L.load().jm_ball_query_ws(B, N, npoint, float(radius), nsample, L.dev(new_xyz, _f32, "new_xyz"),
                                          L.dev(xyz, _f32, "xyz"), L.dev(idx, _i32, "idx"),
                                          ctypes.c_void_p(ws.data_ptr()) if ws is not None else None, ws_bytes, L.stream_ptr()), "ball_query")
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None


ball_query = _BallQuery.apply


def ball_query_dual(radius0: float, nsample0: int, radius1: float, nsample1: int, xyz: torch.Tensor,
                    new_xyz: torch.Tensor, grid: Optional["BallQueryGrid"] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """both MSG radii of an SA level in one pass over xyz (no reference counterpart; results are
    identical to two ball_query calls).  grid: a BallQueryGrid of xyz built ahead (only its query then runs here)"""
    _need(new_xyz, "new_xyz"); _need(xyz, "xyz")
    if grid is not None and grid.matches(xyz) and radius0 > 0 and radius1 > 0:
        return grid.query(new_xyz, radius0, nsample0, radius1, nsample1)
    B, N, _ = xyz.size()
    npoint = new_xyz.size(1)
    idx0 = torch.empty((B, npoint, nsample0), dtype=_i32, device=xyz.device)
    idx1 = torch.empty((B, npoint, nsample1), dtype=_i32, device=xyz.device)
    ws, ws_bytes = _ball_query_workspace(B, N, xyz.device)
    L.check(L.load().jm_ball_query_dual_ws(B, N, npoint, float(radius0), nsample0, float(radius1), nsample1,
                                           L.dev(new_xyz, _f32, "new_xyz"), L.dev(xyz, _f32, "xyz"),
                                           L.dev(idx0, _i32, "idx0"), L.dev(idx1, _i32, "idx1"),
                                           ctypes.c_void_p(ws.data_ptr()) if ws is not None else None, ws_bytes, L.stream_ptr()),
            "ball_query_dual")
    return idx0, idx1


class QueryAndGroup(nn.Module):
    """ball_query + grouping (xyz re-centred on the query point) + feature concat"""

    def __init__(self, radius: float, nsample: int, use_xyz: bool = True):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    def forward(self, xyz: torch.Tensor, new_xyz: torch.Tensor, features: Optional[torch.Tensor] = None,
                idx: Optional[torch.Tensor] = None) -> torch.Tensor:
        """xyz (B, N, 3), new_xyz (B, npoint, 3), features (B, C, N) -> (B, 3 + C, npoint, nsample).
        `idx` may carry a precomputed neighbour list (e.g. from ball_query_dual)."""
        if idx is None:
            idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
        grouped_xyz = grouping_operation(xyz.transpose(1, 2).contiguous(), idx)
        grouped_xyz = grouped_xyz - new_xyz.transpose(1, 2).unsqueeze(-1)
        if features is None:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
            return grouped_xyz
        grouped_features = grouping_operation(features, idx)
        return torch.cat([grouped_xyz, grouped_features], dim=1) if self.use_xyz else grouped_features


class GroupAll(nn.Module):
    """one group containing every point: (B, 3 + C, 1, N)"""

    def __init__(self, use_xyz: bool = True):
        super().__init__()
        self.use_xyz = use_xyz

    def forward(self, xyz: torch.Tensor, new_xyz: Optional[torch.Tensor], features: Optional[torch.Tensor] = None):
        grouped_xyz = xyz.transpose(1, 2).contiguous().unsqueeze(2)
        if features is None:
            return grouped_xyz
        grouped_features = features.unsqueeze(2)
        return torch.cat([grouped_xyz, grouped_features], dim=1) if self.use_xyz else grouped_features
