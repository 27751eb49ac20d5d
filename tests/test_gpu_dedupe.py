"""GPU tier (-m gpu): duplicate-aware set abstraction (csrc/sa_dedupe.hip + jm_sa_mlp_pm_forward_dyn).  The claim is
EXACTNESS: skipping (centre, sample) rows that are exact copies of earlier rows changes nothing — every comparison here is
torch.equal against the dense kernel."""
import numpy as np
import pytest
import torch

from jmodt_amd import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def test_roipool_canonical_count_vs_oracle(oracle):
    """pooled_count = min(points in the enlarged box, S): from the oracle's selected indices (ascending until they wrap)"""
    from jmodt_amd.ops.roipool3d.roipool3d_utils import roipool3d_canonical_gpu
    B, N, M, C, S = 2, 4000, 24, 6, 64
    xyz = synth.kitti_like_cloud(B, N, 5)
    boxes = synth.proposals(xyz, M, 6)
    boxes[0, 0, :3] = 900.0                                            # empty
    boxes[1, 1, 3:6] = 30.0                                            # more than S points inside
    feat = np.random.default_rng(7).normal(size=(B, N, C)).astype(np.float32)
    pooled, empty, count = roipool3d_canonical_gpu(T(xyz), T(feat), T(boxes), 0.2, S, return_count=True)
    wp, we = oracle.roipool3d_canonical(xyz, feat, boxes, 0.2, S)
    _, _, pidx = oracle.roipool3d(xyz, feat, np.stack([oracle.enlarge_box3d(boxes[b], 0.2) for b in range(B)]), S, return_idx=True)
    want = np.zeros((B, M), np.int32)
    for b in range(B):
        for m in range(M):
            if we[b, m]:
                continue
            wrap = np.nonzero(np.diff(pidx[b, m]) <= 0)[0]
            want[b, m] = wrap[0] + 1 if len(wrap) else S
    assert np.array_equal(empty.cpu().numpy(), we) and np.array_equal(count.cpu().numpy(), want)
    assert want[0, 0] == 0 and want[1, 1] == S and 0 < np.median(want) < S
    got = pooled.cpu().numpy()
    assert np.array_equal(got[..., 3:], wp[..., 3:]) and np.abs(got[..., :3] - wp[..., :3]).max() < 1e-5
    cnt = count.cpu().numpy()
    for b in range(B):                                                 # rows count .. S-1 ARE copies of rows 0 .. count-1
        for m in range(M):
            c = cnt[b, m]
            if 0 < c < S:
                assert np.array_equal(got[b, m, c:], got[b, m, np.arange(c, S) % c])


def _padded_sets(R, n, C, seed, max_cnt):
    """R point sets of n points that are cyclic copies of their first cnt points (as roipool3d pads), incl. cnt = 0 (one
    point repeated) and cnt >= n (no copies)"""
    rng = np.random.default_rng(seed)
    cnt = rng.integers(0, max_cnt, R).astype(np.int32)
    cnt[0], cnt[1], cnt[2] = 0, 1, n + 50
    base_xyz = (rng.random((R, n, 3), dtype=np.float32) - 0.5) * np.array([4.0, 1.6, 1.8], np.float32)
    base_f = np.maximum(rng.normal(size=(R, C, n)).astype(np.float32), 0)
    k = np.arange(n)
    xyz, f = np.empty_like(base_xyz), np.empty_like(base_f)
    for r in range(R):
        src = k % max(min(int(cnt[r]), n), 1)
        xyz[r], f[r] = base_xyz[r, src], base_f[r][:, src]
    return xyz, f, np.minimum(cnt, n)


@pytest.mark.parametrize("R,n,max_cnt", [(37, 512, 60), (64, 512, 700), (9, 128, 20)])
def test_sa_dedupe_two_levels_bit_identical_to_dense(R, n, max_cnt):
    """RCNN SA1 / SA2 shapes (config.py:134-139) on cyclically padded sets: dense sa_mlp_pm vs the compacted form, level 2
    fed with level 1's copied centres (canon = level 1's representatives)"""
    from jmodt_amd.ops.pointnet2 import fused, pointnet2_utils as pu
    from jmodt_amd.ops.pointnet2.pointnet2_modules import PointnetSAModule
    torch.manual_seed(R)
    C = 128
    sa1 = PointnetSAModule(mlp=[C, 128, 128, 128], npoint=n // 4, radius=0.2 if n == 512 else 0.5, nsample=64, bn=False).to(DEV).eval()
    sa2 = PointnetSAModule(mlp=[128, 128, 128, 256], npoint=n // 16, radius=0.4 if n == 512 else 1.0, nsample=64, bn=False).to(DEV).eval()
    xyz_np, f_np, cnt = _padded_sets(R, n, C, 3 * R, max_cnt)
    xyz, feats = T(xyz_np), T(f_np)
    g1 = sa1.groupers[0]
    u = fused.hoisted_u_point_major(xyz, feats, sa1.mlps[0])
    assert u is not None and u.shape == (R, n, 128)
    fused.DedupeStats.last.clear()
    canon = fused.canon_from_count(T(cnt), n)
    assert torch.equal(canon[2], torch.arange(n, dtype=torch.int32, device=DEV)) and int(canon[0].max()) == 0
    res = fused.sa_scale_pm_dedupe(xyz, u, sa1.mlps[0], sa1.npoint, g1.radius, g1.nsample, canon, "l1")
    assert res is not None
    new_xyz, out, rep = res
    with torch.no_grad():
        d_xyz, d_out, _ = sa1(xyz, feats)                                # the dense route (fused pm kernel on every row)
    assert torch.equal(new_xyz, d_xyz) and torch.equal(out, d_out)
    # representatives: same coordinates, same output, and rep[rep] == rep
    idx3 = rep.long().unsqueeze(-1).expand(-1, -1, 3)
    assert torch.equal(torch.gather(new_xyz, 1, idx3), new_xyz)
    assert torch.equal(torch.gather(rep, 1, rep.long()), rep)
    assert torch.equal(torch.gather(out, 2, rep.long().unsqueeze(1).expand(-1, out.shape[1], -1)), out)
    # level 2 on level 1's centres
    g2 = sa2.groupers[0]
    u2 = fused.hoisted_u_point_major(new_xyz, out, sa2.mlps[0])
    res2 = fused.sa_scale_pm_dedupe(new_xyz, u2, sa2.mlps[0], sa2.npoint, g2.radius, g2.nsample, rep, "l2")
    assert res2 is not None
    with torch.no_grad():
        d2_xyz, d2_out, _ = sa2(new_xyz, out)
    assert torch.equal(res2[0], d2_xyz) and torch.equal(res2[1], d2_out)
    torch.cuda.synchronize()
    for entry in fused.DedupeStats.last:
        name, dense_rows, counters, cls = entry
        c = counters.cpu().numpy()
        assert c[0] == c[2] and c[1] == (c[0] + 7) // 8 and 0 < c[1] * 128 <= dense_rows + 128 * 8, (name, c, dense_rows)
        # round 4: the segments go through the LISTED kernel — every virtual centre sits in exactly one class, and a segment of
        # d entries runs 2^max(qmin, ceil(log2 d)) rows (quads at SA1's widths, octets at SA2's), never more than the 16 of the
        # segment form
        assert cls is not None and int(cls.sum()) == c[0] and int(cls[:2].sum()) == 0
        assert fused.DedupeStats.rows_executed(entry) <= c[1] * 128
    if max_cnt <= 60:
        entry = fused.DedupeStats.last[0]
        assert fused.DedupeStats.rows_executed(entry) < entry[1] // 8    # few distinct points: >= 8x fewer rows
    # the same scales through the 16-row segment form (round 3): identical bits
    fused.DEDUPE_LISTED = False
    try:
        seg = fused.sa_scale_pm_dedupe(xyz, u, sa1.mlps[0], sa1.npoint, g1.radius, g1.nsample, canon, "l1")
        assert fused.DedupeStats.last[-1][3] is None
    finally:
        fused.DEDUPE_LISTED = True
    assert torch.equal(seg[1], out)


@pytest.mark.parametrize("kind", ["uniform", "kitti"])
def test_engine_dedupe_on_off_bit_identical(kind):
    """the benchmarked engine (full widths, 16384-point frames, 128 RoIs x 512 points) with and without the compaction"""
    from jmodt_amd.detector import DetectorConfig
    from jmodt_amd.ops.pointnet2 import fused
    from tests.test_gpu_detector import make_engine
    eng = make_engine(seed=5, cfg=DetectorConfig.survey()).to(DEV)
    xyz, img, xy = synth.frames(2, 16384, 99, kind=kind)
    a = [T(xyz), T(img), T(xy)]
    with torch.no_grad():
        eng.dedupe_rcnn = True
        c1, aff1, i1 = eng(*a)
        stats = list(fused.DedupeStats.last)
        eng.dedupe_rcnn = False
        dense = eng.rcnn_forward(i1["pts_input"])                       # the same pooled RoIs through the dense kernels
        assert not fused.DedupeStats.last
        eng.dedupe_rcnn = True
        again = eng.rcnn_forward(i1["pts_input"])                       # (count kept from the forward: compacted again)
    torch.cuda.synchronize()
    assert [s[0] for s in stats] == ["rcnn_sa1", "rcnn_sa2"]
    for k in ("rcnn_feat", "rcnn_cls", "rcnn_reg"):
        assert torch.equal(i1[k], dense[k]), k
        assert torch.equal(again[k], dense[k]), k
    rows = {e[0]: (e[1], fused.DedupeStats.rows_executed(e)) for e in stats}
    assert rows["rcnn_sa1"][1] < rows["rcnn_sa1"][0] and rows["rcnn_sa2"][1] < rows["rcnn_sa2"][0]
    print(kind, "rows dense -> executed:", rows)
