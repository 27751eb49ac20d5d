"""GPU tier (-m gpu): the duplicate-aware (LISTED) form of the RPN set-abstraction scales (csrc/sa_groups.hip + the listed modes
of the fused SA kernels).  ball_query back-fills a short neighbour list with its first hit (ball_query_gpu.cu:36-40) and the
max-pool of pointnet2_modules.py:50-52 is idempotent, so only a group's first d rows matter; the listed form executes
2^ceil(log2 d) of them.  The claim is EXACTNESS: every comparison is torch.equal against the dense kernel of the same scale."""
import numpy as np
import pytest
import torch

from jmodt_amd import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _lists(B, M, N, ns, pattern, seed):
    """(B, M, ns) int32 neighbour lists in ball_query's form: d ascending distinct point indices, then copies of the first"""
    rng = np.random.default_rng(seed)
    idx = np.zeros((B, M, ns), np.int32)
    d = {"singletons": np.ones((B, M), np.int64), "full": np.full((B, M), ns),
         "mixed": rng.integers(1, ns + 1, (B, M)), "sparse": np.minimum(rng.geometric(0.6, (B, M)), ns)}[pattern]
    for b in range(B):
        for m in range(M):
            k = int(d[b, m])
            hits = np.sort(rng.choice(N, k, replace=False))
            idx[b, m, :k] = hits
            idx[b, m, k:] = hits[0]
    if pattern == "mixed":
        idx[0, 0] = 0                                   # a centre without any hit keeps the caller's zero fill
        idx[0, 1] = np.array([5, 5, 5, 7] + [5] * (ns - 4))     # NOT ball_query's form: the last differing slot decides
    return idx, d


@pytest.mark.parametrize("ns", [16, 32])
@pytest.mark.parametrize("qmin", [0, 2])
def test_group_plan_bins_every_group_once_into_its_class(ns, qmin):
    from jmodt_amd.ops.pointnet2.fused import group_plan
    B, M, N = 3, 700, 900
    idx, _ = _lists(B, M, N, ns, "mixed", 3)
    cnt_t, gl_t = group_plan(T(idx), qmin)
    plan = torch.cat([cnt_t, gl_t]).cpu().numpy()
    G = B * M
    flat = idx.reshape(G, ns)
    need = np.array([1 + max([s for s in range(ns) if row[s] != row[0]], default=0) for row in flat])
    q = np.maximum(np.ceil(np.log2(np.maximum(need, 1))).astype(int), qmin)
    nq = int(np.log2(ns)) + 1
    assert plan.shape[0] == 8 + nq * G
    seen = []
    for c in range(8):
        cnt = plan[c]
        assert cnt == (q == c).sum(), (c, cnt)
        if c < nq:
            members = plan[8 + c * G: 8 + c * G + cnt]
            assert np.array_equal(np.sort(members), np.nonzero(q == c)[0])
            seen.append(members)
    assert np.array_equal(np.sort(np.concatenate(seen)), np.arange(G))


WIDE = [  # (C, mlp spec after the +3, M, N): RPN SA3 / SA4 scales (config.py:75-82) and a narrow odd one
    (256, [128, 196, 256], 256, 1024), (512, [256, 256, 512], 64, 256), (512, [256, 384, 512], 64, 256), (40, [48, 72], 98, 300)]


@pytest.mark.parametrize("C,spec,M,N", WIDE)
@pytest.mark.parametrize("ns", [16, 32])
@pytest.mark.parametrize("pattern", ["singletons", "full", "mixed", "sparse"])
def test_listed_wide_kernel_is_bit_identical_to_the_dense_kernel(C, spec, M, N, ns, pattern):
    """sa_mlp_wide_kernel: tiles of one class each (32 >> q groups, pool over 2^q rows) == one group (or two) per tile"""
    from jmodt_amd.ops.pointnet2 import fused
    from jmodt_amd.ops.pointnet2.pytorch_utils import SharedMLP
    B = 4
    torch.manual_seed(C + ns)
    mlp = SharedMLP([C + 3] + spec, bn=True).to(DEV).eval()
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for m in mlp.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
    xyz = synth.dense_cloud(B, N, 7, extent=6.0)
    idx, d = _lists(B, M, N, ns, pattern, 11)
    new_xyz = np.take_along_axis(xyz, idx[:, :, :1].astype(np.int64).repeat(3, 2), 1) + np.float32(0.01)
    feats = np.random.default_rng(5).normal(size=(B, C, N)).astype(np.float32)
    args = (T(xyz), T(new_xyz), T(feats), T(idx), mlp)
    assert fused.listed_kind(mlp, args[2], args[3], B, N) == 2
    full = torch.zeros((B, spec[-1] + 8, M), device=DEV)
    dense = fused.sa_mlp_fused(*args, listed=False)
    listed = fused.sa_mlp_fused(*args, listed=True)
    into = fused.sa_mlp_fused(*args, out=full[:, 3:3 + spec[-1]], listed=True)        # a channel slice of a wider tensor
    assert torch.equal(listed, dense)
    assert torch.equal(into, dense) and float(full[:, :3].abs().max()) == 0 and float(full[:, 3 + spec[-1]:].abs().max()) == 0
    # the rows the listed form executed: 2^ceil(log2 d) per group, from the plan's class counts
    plan = fused.ListedStats.last[-1][3].cpu().numpy()
    rows = sum(int(plan[c]) << c for c in range(8))
    assert rows <= 2 * int(d.sum()) + 4 and rows <= B * M * ns and int(plan.sum()) == B * M
    if pattern == "singletons":
        assert rows == B * M
    if pattern == "full":
        assert rows == B * M * ns


@pytest.mark.parametrize("spec,ns", [([16, 16, 32], 16), ([32, 32, 64], 32)])
@pytest.mark.parametrize("pattern", ["singletons", "full", "mixed", "sparse"])
def test_listed_xyz_kernel_is_bit_identical_to_the_dense_kernel(spec, ns, pattern):
    """sa_xyz_valu_kernel (the xyz-only scales of the first RPN level, config.py:75-82): passes of 1024 rows of one class"""
    from jmodt_amd.ops.pointnet2 import fused
    from jmodt_amd.ops.pointnet2.pytorch_utils import SharedMLP
    B, M, N = 3, 320, 2000
    torch.manual_seed(ns)
    mlp = SharedMLP([3] + spec, bn=True).to(DEV).eval()
    g = torch.Generator().manual_seed(2)
    with torch.no_grad():
        for m in mlp.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
    xyz = synth.dense_cloud(B, N, 9, extent=3.0)
    idx, d = _lists(B, M, N, ns, pattern, 13)
    new_xyz = np.take_along_axis(xyz, idx[:, :, :1].astype(np.int64).repeat(3, 2), 1)
    args = (T(xyz), T(new_xyz), None, T(idx), mlp)
    assert fused.listed_kind(mlp, None, args[3], B, N) == 3
    dense = fused.sa_mlp_fused(*args, listed=False)
    listed = fused.sa_mlp_fused(*args, listed=True)
    full = torch.zeros((B, spec[-1] + 5, M), device=DEV)
    into = fused.sa_mlp_fused(*args, out=full[:, 2:2 + spec[-1]], listed=True)
    assert torch.equal(listed, dense) and torch.equal(into, dense)
    assert float(full[:, :2].abs().max()) == 0 and float(full[:, 2 + spec[-1]:].abs().max()) == 0
    assert float(dense.abs().max()) > 0


@pytest.mark.parametrize("C,spec,ns", [(96, [64, 64, 128], 16), (96, [64, 96, 128], 32), (29, [32, 48, 96], 16),
                                       (128, [128, 128, 128], 64), (128, [128, 128, 256], 64)])     # the RCNN scales (ns = 64, LDS-tight)
@pytest.mark.parametrize("pattern", ["singletons", "full", "mixed", "sparse"])
def test_listed_pm_kernel_is_bit_identical_to_the_dense_kernel(C, spec, ns, pattern):
    """sa_mlp_pm_kernel (pre-projected two-layer scales: RPN SA2, config.py:75-82): 128-row tiles of one class each, classes of
    4 .. nsample rows (the accumulator layout pools four consecutive rows inside a lane), partial maxima per row quad"""
    from jmodt_amd.ops.pointnet2 import fused
    from jmodt_amd.ops.pointnet2.pytorch_utils import SharedMLP
    B, M, N = 3, 264, 1024
    torch.manual_seed(ns + C)
    mlp = SharedMLP([C + 3] + spec, bn=True).to(DEV).eval()
    g = torch.Generator().manual_seed(2)
    with torch.no_grad():
        for m in mlp.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
    xyz = synth.dense_cloud(B, N, 9, extent=3.0)
    idx, d = _lists(B, M, N, ns, pattern, 17)
    new_xyz = np.take_along_axis(xyz, idx[:, :, :1].astype(np.int64).repeat(3, 2), 1) + np.float32(0.02)
    feats = np.random.default_rng(6).normal(size=(B, C, N)).astype(np.float32)
    args = (T(xyz), T(new_xyz), T(feats), T(idx), mlp)
    assert fused.pm_plan(mlp, DEV, B, N, M, ns) is not None
    fused.ListedStats.last.clear()
    dense = fused.sa_mlp_fused(*args, listed=False)
    assert not fused.ListedStats.last
    listed = fused.sa_mlp_fused(*args, listed=True)
    assert fused.ListedStats.last and fused.ListedStats.last[-1][0].endswith("sa_mlp_pm_forward_listed")
    full = torch.zeros((B, spec[-1] + 5, M), device=DEV)
    into = fused.sa_mlp_fused(*args, out=full[:, 2:2 + spec[-1]], listed=True)
    assert torch.equal(listed, dense) and torch.equal(into, dense)
    assert float(full[:, :2].abs().max()) == 0 and float(full[:, 2 + spec[-1]:].abs().max()) == 0
    plan = fused.ListedStats.last[-1][3].cpu().numpy()
    rows = sum(int(plan[c]) << c for c in range(8))
    from jmodt_amd import _lib
    qmin = int(_lib.load().jm_sa_mlp_pm_listed_qmin(spec[0], spec[1], spec[2]))
    assert qmin == 2                                            # quads at every shape of the reference configuration
    assert int(plan[:qmin].sum()) == 0 and rows <= B * M * ns
    if pattern == "singletons":
        assert rows == (1 << qmin) * B * M


@pytest.mark.parametrize("kind", ["uniform", "kitti", "packed"])
def test_rpn_levels_listed_on_off_bit_identical_at_full_width(kind):
    """the four set-abstraction levels of the BENCHMARKED backbone (DetectorConfig.survey(): 16384 -> 4096 -> 1024 -> 256 -> 64
    centres, both scales per level, full widths) on the three synthetic clouds: the listed form (sa_xyz_valu / sa_mlp_pm /
    sa_mlp_wide in listed mode) against the dense kernels, level by level on identical inputs, torch.equal; and the rows it
    executed, from the plans' class counts (94-97 % of the dense rows are copies on the headline cloud, none on the packed one)"""
    from jmodt_amd.detector import DetectorConfig
    from jmodt_amd.ops.pointnet2 import fused, pointnet2_utils
    from tests.test_gpu_detector import make_engine
    eng = make_engine(seed=5, cfg=DetectorConfig.survey()).to(DEV).eval()
    net = eng.rpn.backbone_net
    xyz = T(synth.frames(2, 16384, 99, kind=kind)[0])
    g = torch.Generator().manual_seed(3)
    cur, feats = xyz, None
    executed, dense_rows = 0, 0
    kernels = []
    with torch.no_grad():
        for lv, sa in enumerate(net.SA_modules):
            _, new_xyz = pointnet2_utils.farthest_point_sample_xyz(cur, sa.npoint)
            fused.LISTED = False
            try:
                _, want, _ = sa(cur, feats, new_xyz=new_xyz)
            finally:
                fused.LISTED = True
            fused.ListedStats.last.clear()
            _, got, _ = sa(cur, feats, new_xyz=new_xyz)
            assert torch.equal(got, want), (kind, lv)
            assert len(fused.ListedStats.last) == 2, (kind, lv, fused.ListedStats.last)          # both scales took the listed form
            for name, rows, ns, plan in fused.ListedStats.last:
                pl = plan[:8].cpu().numpy()
                assert int(pl.sum()) == cur.shape[0] * sa.npoint                                # every group in exactly one class
                executed += sum(int(pl[c]) << c for c in range(8))
                dense_rows += rows
                kernels.append(name)
            cur = new_xyz
            feats = torch.relu(torch.randn(cur.shape[0], got.shape[1], cur.shape[1], generator=g)).to(DEV)   # the next level's input
    assert [k.split("_forward")[0] for k in kernels] == ["sa_mlp", "sa_mlp", "sa_mlp_pm", "sa_mlp_pm", "sa_mlp", "sa_mlp", "sa_mlp", "sa_mlp"]
    print(kind, "rows dense -> executed:", dense_rows, executed, round(executed / dense_rows, 4))
    if kind == "uniform":
        assert executed < 0.12 * dense_rows
    elif kind == "kitti":
        assert executed < 0.25 * dense_rows
    else:
        assert executed <= dense_rows


def test_group_plan_dual_equals_two_single_plans():
    """the two scales of a level planned by one launch: the same class counts and the same class members as two single plans"""
    from jmodt_amd.ops.pointnet2.fused import group_plan, group_plan_dual
    B, M, N = 2, 900, 1200
    i0, _ = _lists(B, M, N, 16, "sparse", 1)
    i1, _ = _lists(B, M, N, 32, "mixed", 2)
    (c0, g0), (c1, g1) = group_plan_dual(T(i0), 0, T(i1), 2)
    for (c, g), (idx, q, ns) in zip(((c0, g0), (c1, g1)), ((i0, 0, 16), (i1, 2, 32))):
        cs, gs = group_plan(T(idx), q)
        assert torch.equal(c, cs)
        G = B * M
        for k in range(int(np.log2(ns)) + 1):
            n = int(c[k])
            assert torch.equal(torch.sort(g[k * G:k * G + n])[0], torch.sort(gs[k * G:k * G + n])[0])


def test_group_plan_edge_cases():
    """no groups at all; a device-side group count below the capacity (the rest of idx is never read, even when it is garbage);
    a capacity that is no multiple of the plan kernel's workgroup"""
    from jmodt_amd.ops.pointnet2.fused import group_plan
    cnt, gl = group_plan(torch.empty((0, 5, 16), dtype=torch.int32, device=DEV), 0)
    assert int(cnt.sum()) == 0
    idx, _ = _lists(1, 777, 500, 16, "mixed", 9)
    t = T(idx)
    t[0, 300:] = 0x7fffffff                                          # garbage behind the valid prefix
    valid = torch.tensor([300], dtype=torch.int32, device=DEV)
    cnt, gl = group_plan(t, 2, valid)
    ref_cnt, ref_gl = group_plan(T(idx[:, :300]), 2)
    assert torch.equal(cnt, ref_cnt) and int(cnt.sum()) == 300
    for q in range(5):
        n = int(cnt[q])
        assert torch.equal(torch.sort(gl[q * 777:q * 777 + n])[0], torch.sort(ref_gl[q * 300:q * 300 + n])[0])
