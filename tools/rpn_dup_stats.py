"""how many DISTINCT points do the ball-query groups of the RPN set-abstraction levels hold?  (CPU oracle: FPS chain + ball query on one
synthetic frame per cloud kind) -> the rows a duplicate-aware form of the level's MLP would execute, as for RCNN SA1 / SA2"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jmodt_amd import synth
from oracle import oracle as O

npoints, radii, nsamples = (4096, 1024, 256, 64), ((0.1, 0.5), (0.5, 1.0), (1.0, 2.0), (2.0, 4.0)), (16, 32)
for kind in ("uniform", "kitti", "packed"):
    xyz = synth.frames(1, 16384, 11, kind=kind)[0].astype(np.float32)
    cur = xyz
    line = []
    for lv, m in enumerate(npoints):
        idx = O.furthest_point_sample(cur, m)
        new = np.take_along_axis(cur, idx[..., None].astype(np.int64), axis=1)
        for r, ns in zip(radii[lv], nsamples):
            nb = O.ball_query(r, ns, cur, new)[0]                         # (m, ns)
            distinct_idx = np.array([len(set(row.tolist())) for row in nb])
            pts = cur[0][nb]                                              # (m, ns, 3): distinct COORDINATES (duplicate points count once)
            distinct_xyz = np.array([len({tuple(p) for p in g.tolist()}) for g in pts])
            line.append(f"L{lv + 1} r={r}: {distinct_xyz.mean():5.2f} / {ns} distinct ({distinct_xyz.sum() / (m * ns):.3f} of the rows; by index {distinct_idx.mean():.2f})")
        cur = new
    print(kind)
    for l in line: print("   ", l)
