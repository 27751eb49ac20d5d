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

`PointnetSAModuleMSG.forward` already takes `new_xyz` (pointnet2_modules.py:24-33).
"""
from typing import List, Tuple

import torch

from . import pointnet2_utils

_side = {}


def _side_stream(device) -> torch.cuda.Stream:
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    if key not in _side:
        _side[key] = torch.cuda.Stream(device=key)
    return _side[key]


class FpsPyramid:
    def __init__(self, xyz: torch.Tensor, npoints: List[int]):
        main = torch.cuda.current_stream(xyz.device)
        side = _side_stream(xyz.device)
        side.wait_stream(main)           # xyz is ready; previous consumers of our buffers are done
        self._levels: List[Tuple[torch.Tensor, torch.Tensor, torch.cuda.Event]] = []
        with torch.cuda.stream(side):
            cur = xyz
            for m in npoints:
                idx, new_xyz = pointnet2_utils.farthest_point_sample_xyz(cur, m)
                ev = torch.cuda.Event()
                ev.record(side)
                # handed to the main stream: keep the allocator from recycling them under it
                idx.record_stream(main)
                new_xyz.record_stream(main)
                self._levels.append((idx, new_xyz, ev))
                cur = new_xyz

    def level(self, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """(idx (B, m_k) int32, new_xyz (B, m_k, 3)) of level k, ordered after its FPS on the current stream"""
        idx, new_xyz, ev = self._levels[k]
        torch.cuda.current_stream(idx.device).wait_event(ev)
        return idx, new_xyz
