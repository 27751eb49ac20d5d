"""FPS for every set-abstraction level, ahead of the levels themselves.

Furthest point sampling depends on coordinates only (pointnet2_modules.py:35-39: `new_xyz` is
gathered from `xyz` by the FPS indices; features never enter), so the chain
16384 -> 4096 -> 1024 -> 256 -> 64 of the RPN backbone (config.py:75) can run on its own HIP
stream while the main stream does neighbour search, grouping, MLPs and LI-Fusion of the earlier
levels.  FPS is one workgroup per cloud — 8 of 256 CUs at batch 8 — so the two streams do not
compete for the machine.

    pyr = FpsPyramid(xyz, [4096, 1024, 256, 64])
    for k, sa in enumerate(sa_modules):
        idx, new_xyz = pyr.level(k)                    # main stream waits for level k only
        xyz_k, feats_k, _ = sa(xyz_k, feats_k, new_xyz=new_xyz)
    pyr.release()

`PointnetSAModuleMSG.forward` already takes `new_xyz` (pointnet2_modules.py:24-33).
"""
from typing import List, Optional, Tuple

import torch

from ...profile import prof
from . import pointnet2_utils

_side = {}


SIDE_PRIORITY = {}     # slot -> stream priority (default 0 everywhere; tools/stream_prio_probe.py fills it — no environment switch)


def _side_priority(slot: int) -> int:
    return int(SIDE_PRIORITY.get(slot, 0))


def side_stream(device, slot: int = 0) -> torch.cuda.Stream:
    """a per-(device, slot) auxiliary stream, created once (slot 0: FPS chain, 1: image branch, 2: start/end head, 3: detections,
    4-6: further FPS chains in flight, detector.prefetch_depth > 1)"""
    d = torch.device(device)
    key = (d.index if d.index is not None else torch.cuda.current_device(), slot)
    if key not in _side:
        _side[key] = torch.cuda.Stream(device=key[0], priority=_side_priority(slot))
    return _side[key]


_side_stream = side_stream   # round-1 name


class FpsPyramid:
    def __init__(self, xyz: torch.Tensor, npoints: List[int], overlap: bool = True, with_interp: bool = False,
                 grid_radii: Optional[List[float]] = None, slot: int = 0):
        """slot: which side stream (`side_stream(device, slot)`) — several pyramids in flight need one each.
        grid_radii[k] (optional): the largest ball-query radius of level k — the level's neighbour-search grid
        (pointnet2_utils.BallQueryGrid over the level's INPUT points) is then built here, on the side stream, ahead of the
        sampling that produces the level's centres; `grid(k)` hands it out"""
        main = torch.cuda.current_stream(xyz.device)
        side = side_stream(xyz.device, slot) if overlap else main
        self._main, self._side, self._xyz = main, side, xyz   # xyz stays referenced until release()
        side.wait_stream(main)           # xyz is ready; previous consumers of our buffers are done
        if side is not main:
            xyz.record_stream(side)      # allocated on the main stream, read by the side stream
        self._levels: List[Tuple[torch.Tensor, torch.Tensor, torch.cuda.Event]] = []
        self._grids: List[Optional[pointnet2_utils.BallQueryGrid]] = []
        with torch.cuda.stream(side), prof.scope("fps_pyramid"):
            cur = xyz
            for k, m in enumerate(npoints):
                with prof.scope(f"L{k + 1}"):
                    g = None
                    if grid_radii is not None and k < len(grid_radii) and grid_radii[k]:
                        g = pointnet2_utils.BallQueryGrid(cur, grid_radii[k])      # (points + radius only: in front of the sampling)
                        g = g if g.ws is not None else None
                    self._grids.append(g)
                    idx, new_xyz = pointnet2_utils.farthest_point_sample_xyz(cur, m)
                ev = torch.cuda.Event()
                ev.record(side)
                # handed to the announcing stream WITHOUT record_stream: the tensors stay referenced until release(), which
                # orders the side stream after the consumer before dropping them — a record_stream would make the caching
                # allocator record one event per block on the consumer's stream at that point (a train of markers in the
                # middle of its kernel chain, ~100 us per pyramid: tools/timeline.sh)
                self._levels.append((idx, new_xyz, ev))
                cur = new_xyz
            # the feature-propagation modules' neighbour search (three_nn + inverse-distance weights,
            # pointnet2_modules.py:147-152) depends on coordinates only as well: level k-1 <- level k, behind the chain
            self._interp: List[Tuple[torch.Tensor, torch.Tensor, torch.cuda.Event]] = []
            if with_interp:
                with prof.scope("fp_neighbours"):
                    for k in range(len(npoints)):
                        unknown = xyz if k == 0 else self._levels[k - 1][1]
                        nn3, w = pointnet2_utils.three_nn_weights(unknown, self._levels[k][1])
                        ev = torch.cuda.Event()
                        ev.record(side)
                        self._interp.append((nn3, w, ev))

    def level(self, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """(idx (B, m_k) int32, new_xyz (B, m_k, 3)) of level k, ordered after its FPS on the current stream"""
        idx, new_xyz, ev = self._levels[k]
        cur = torch.cuda.current_stream(idx.device)
        if cur is not self._side and cur != self._side:
            if cur != self._main:           # consumed on another stream than the one that announced the cloud
                idx.record_stream(cur)
                new_xyz.record_stream(cur)
            # how long the consumer is actually held up = the EXPOSED part of this level's sampling
            prof.stall(f"fps_exposed_wait_L{k + 1}", lambda: cur.wait_event(ev))
        return idx, new_xyz

    def grid(self, k: int):
        """the BallQueryGrid over level k's input points (None: not requested / the scan is used at that size); ordered like
        level(k): call it after level(k)"""
        return self._grids[k] if k < len(self._grids) else None

    def interp(self, k: int):
        """(nn3 (B, n_{k-1}, 3) int32, weights (B, n_{k-1}, 3)) interpolating level k's features onto level k-1's points
        (level -1 = the input cloud), or None when the pyramid was built without them"""
        if not self._interp:
            return None
        nn3, w, ev = self._interp[k]
        cur = torch.cuda.current_stream(nn3.device)
        if cur != self._side:
            if cur != self._main:
                nn3.record_stream(cur)
                w.record_stream(cur)
            cur.wait_event(ev)
        return nn3, w

    def release(self):
        """the consumer is done with every level: order the side stream after it, drop the references"""
        if self._side is not self._main:
            self._side.wait_stream(torch.cuda.current_stream(self._xyz.device))
        self._levels, self._interp, self._xyz, self._grids = [], [], None, []
