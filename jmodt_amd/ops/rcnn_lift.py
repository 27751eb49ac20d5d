"""The RCNN stage's per-point input MLP (xyz_up_layer + merge_down_layer, jmodt/detection/modeling/rcnn.py:176-184)
as one fp32-MFMA kernel on the pooled RoI points (csrc/rcnn_lift.hip), optionally with the first set-abstraction
layer hoisted in front of its gather (its output is then `u` for ops/pointnet2/fused.sa_mlp_pre_from_u)."""
import ctypes
from typing import Optional, Sequence, Tuple

import torch

from .. import _lib as L
from .fusion import _pack

_f32 = torch.float32


class PackedRcnnLift:
    def __init__(self, up: Sequence[Tuple[torch.Tensor, torch.Tensor]], merge: Tuple[torch.Tensor, torch.Tensor],
                 hoist: Optional[Tuple[torch.Tensor, torch.Tensor]] = None):
        """up = [(W (h1, K), b), (W (h2, h1), b)] xyz_up_layer; merge = (W (hm, h2 + C), b) merge_down_layer;
        hoist = (W1 (ho, 3 + hm), b1): the first SA layer's folded weight in QueryAndGroup's [xyz | features] order"""
        (Wu1, bu1), (Wu2, bu2) = up
        Wm, bm = merge
        self.K, self.h1, self.h2, self.hm = Wu1.shape[1], Wu1.shape[0], Wu2.shape[0], Wm.shape[0]
        self.C = Wm.shape[1] - self.h2
        self.ho = 0
        self.wu1, self.bu1 = _pack(Wu1, bu1)
        self.wu2, self.bu2 = _pack(Wu2, bu2)
        self.wmh, self.bm = _pack(Wm[:, :self.h2], bm)
        self.wmf, _ = _pack(Wm[:, self.h2:], None)
        self.wom = self.wox = self.bo = None
        if hoist is not None:
            W1, b1 = hoist
            assert W1.shape[1] == 3 + self.hm
            self.ho = W1.shape[0]
            self.wom, self.bo = _pack(W1[:, 3:], b1)
            wx = torch.zeros((self.ho, self.K), dtype=_f32, device=W1.device)
            wx[:, :3] = W1[:, :3]
            self.wox, _ = _pack(wx, None)

    def supported(self, S: int) -> bool:
        return bool(L.load().jm_rcnn_lift_supported(S, self.K, self.C, self.h1, self.h2, self.hm, self.ho))

    @torch.no_grad()
    def __call__(self, pts_input: torch.Tensor, point_major: bool = False, count: Optional[torch.Tensor] = None) -> torch.Tensor:
        """pts_input (R, S, K + C) contiguous -> (R, hm, S) merged features, or (R, ho, S) = u when a layer is hoisted;
        point_major: (R, S, h) instead (the layout the point-major set-abstraction kernel gathers).
        count (R,) int32: distinct points per slab (rows count .. S-1 are cyclic copies): 32-point tiles of pure copies are
        skipped and their output rows stay UNINITIALISED — only for consumers that read canonical rows (fused.sa_scale_pm_dedupe)"""
        p = pts_input.to(_f32).contiguous()
        R, S, _ = p.shape
        h = self.ho or self.hm
        out = torch.empty((R, S, h) if point_major else (R, h, S), dtype=_f32, device=p.device)

        def ptr(t):
            return L.dev(t, _f32, "w") if t is not None else None
        if count is not None:
            work = torch.empty((1 + R * (S // 32),), dtype=torch.int32, device=p.device)
            L.check(L.load().jm_rcnn_lift_forward_cnt(R, S, self.K, self.C, self.h1, self.h2, self.hm, self.ho, L.dev(p, _f32, "pts_input"),
                                                      ptr(self.wu1), ptr(self.bu1), ptr(self.wu2), ptr(self.bu2), ptr(self.wmh),
                                                      ptr(self.wmf), ptr(self.bm), ptr(self.wom), ptr(self.wox), ptr(self.bo),
                                                      int(point_major), ctypes.c_void_p(out.data_ptr()),
                                                      L.dev(count.contiguous(), torch.int32, "count"),
                                                      L.dev(work, torch.int32, "work"), L.stream_ptr()), "rcnn_lift")
            return out
        L.check(L.load().jm_rcnn_lift_forward(R, S, self.K, self.C, self.h1, self.h2, self.hm, self.ho, L.dev(p, _f32, "pts_input"),
                                              ptr(self.wu1), ptr(self.bu1), ptr(self.wu2), ptr(self.bu2), ptr(self.wmh),
                                              ptr(self.wmf), ptr(self.bm), ptr(self.wom), ptr(self.wox), ptr(self.bo), int(point_major),
                                              ctypes.c_void_p(out.data_ptr()), L.stream_ptr()), "rcnn_lift")
        return out
