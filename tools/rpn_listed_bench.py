"""dense vs LISTED form of the RPN set-abstraction scales, isolated (HIP events, 30 repetitions), per level and list pattern:
singletons = the headline (uniform) cloud's levels 2-4, sparse = KITTI-like (1-3 distinct), full = packed (no back-fill).
    gpurun -- 'python tools/rpn_listed_bench.py'"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from jmodt_amd import synth
from jmodt_amd.ops.pointnet2 import fused
from jmodt_amd.ops.pointnet2.pytorch_utils import SharedMLP
from test_gpu_listed import _lists, T

DEV = "cuda:0"
LEVELS = [("L1", 0, [[16, 16, 32], [32, 32, 64]], 4096, 16384), ("L2", 96, [[64, 64, 128], [64, 96, 128]], 1024, 4096),
          ("L3", 256, [[128, 196, 256], [128, 196, 256]], 256, 1024), ("L4", 512, [[256, 256, 512], [256, 384, 512]], 64, 256)]
only = sys.argv[1:] or [l[0] for l in LEVELS]


from rpn_listed_bench_timeit import timeit


B = 8
for name, C, specs, M, N in LEVELS:
    if name not in only:
        continue
    xyz = synth.dense_cloud(B, N, 7, extent=6.0)
    feats = T(np.random.default_rng(5).normal(size=(B, C, N)).astype(np.float32)) if C else None
    for spec, ns in zip(specs, (16, 32)):
        torch.manual_seed(1)
        mlp = SharedMLP([C + 3] + spec, bn=True).to(DEV).eval()
        for pattern in ("singletons", "sparse", "full"):
            idx, d = _lists(B, M, N, ns, pattern, 11)
            new_xyz = np.take_along_axis(xyz, idx[:, :, :1].astype(np.int64).repeat(3, 2), 1)
            args = (T(xyz), T(new_xyz), feats, T(idx), mlp)
            kind = fused.listed_kind(mlp, feats, args[3], B, N)
            td = timeit(lambda: fused.sa_mlp_fused(*args, listed=False))
            fused.ListedStats.last.clear()
            tl = timeit(lambda: fused.sa_mlp_fused(*args, listed=True))
            kind = kind or ("pm" if fused.ListedStats.last else 0)
            same = torch.equal(fused.sa_mlp_fused(*args, listed=False), fused.sa_mlp_fused(*args, listed=True))
            tp = timeit(lambda: fused.group_plan(args[3], 0))
            print(f"{name} ns={ns:2d} {str(spec):<16} {pattern:<10} mean d {d.mean():5.2f}: dense {td:7.1f} us  listed {tl:7.1f} us "
                  f"(plan alone {tp:5.1f})  kind {kind}  equal {same}", flush=True)
