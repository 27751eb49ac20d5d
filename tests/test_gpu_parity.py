"""GPU tier (-m gpu): the HIP kernels, called through the Python operator API -> ctypes -> C ABI
(include/jmodt_hip.h), against the CPU oracle on the same seeded inputs, against the committed
golden vectors, and — at BASELINE.json's full sizes — through size-independent properties.

Bars: bit-exact for every index / integer output (FPS, ball query, three_nn indices, roipool
indices and copied features, NMS keep lists, NMS masks); 1e-4 absolute for float results
(interpolation, BEV overlap, gather, affinity scores), the tolerance BASELINE.json states.
"""
import numpy as np
import pytest
import torch

from jmodt_amd import synth
from tests.conftest import load_golden

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.fixture(scope="module", autouse=True)
def _require_gpu_and_lib():
    assert torch.cuda.is_available(), "GPU tier needs a GPU"
    from jmodt_amd import _lib
    _lib.load()


# ------------------------------------------------------------------ FPS
@pytest.mark.parametrize("B,N,m,kw", [
    (2, 1024, 256, {}), (2, 1000, 200, {}), (1, 256, 64, dict(dup_frac=0.3)),
    (2, 512, 128, dict(quantize=2.0 ** -3)), (3, 4096, 512, dict(dup_frac=0.1)),
    (1, 64, 16, {}), (2, 40, 12, {}), (1, 1, 1, {}), (1, 5, 5, {}), (2, 2048, 300, {}), (1, 8192, 256, {}),
    (64, 512, 128, {}), (64, 128, 32, {}),       # RCNN-stage shapes (config.py:134), many clouds
])
def test_fps_bit_exact(oracle, B, N, m, kw):
    from jmodt_amd.ops.pointnet2.pointnet2_utils import farthest_point_sample
    xyz = synth.cloud(B, N, seed=11, **kw)
    got = farthest_point_sample(T(xyz), m).cpu().numpy()
    assert got.dtype == np.int32
    assert np.array_equal(got, oracle.furthest_point_sample(xyz, m))


@pytest.mark.parametrize("B,N,m", [(2, 1000, 100), (2, 4096, 300), (1, 16384, 200), (3, 256, 64), (40, 512, 128),
                                   (2, 40000, 120)])
def test_fps_extension_level_temp_buffer(oracle, B, N, m):
    """pointnet2_cuda.farthest_point_sampling_wrapper also leaves `temp` (the caller's 1e10-filled scratch,
    pointnet2_utils.py:25-27) exactly as the reference kernel does: the running min-distances after m-1
    updates — checked for the single-wave, multi-wave and co-operative kernels"""
    from jmodt_amd.ext import pointnet2_cuda
    xyz = synth.cloud(B, N, seed=N + m, dup_frac=0.05)
    t = T(xyz)
    temp = torch.full((B, N), 1e10, dtype=torch.float32, device=DEV)
    idx = torch.zeros((B, m), dtype=torch.int32, device=DEV)
    assert pointnet2_cuda.farthest_point_sampling_wrapper(B, N, m, t, temp, idx) == 1
    want_idx, want_temp = oracle.furthest_point_sample(xyz, m, return_temp=True)
    assert np.array_equal(idx.cpu().numpy(), want_idx)
    assert np.array_equal(temp.cpu().numpy(), want_temp)


def test_fps_with_coordinates_in_one_call(oracle):
    from jmodt_amd.ops.pointnet2.pointnet2_utils import farthest_point_sample_xyz
    for B, N, m in ((3, 3000, 257), (2, 40000, 100), (70, 512, 128)):
        xyz = synth.cloud(B, N, seed=B + N)
        idx, new_xyz = farthest_point_sample_xyz(T(xyz), m)
        want = oracle.furthest_point_sample(xyz, m)
        assert np.array_equal(idx.cpu().numpy(), want)
        assert np.array_equal(new_xyz.cpu().numpy(), np.take_along_axis(xyz, want[..., None].astype(np.int64), axis=1))


def test_fps_all_equal_and_golden(oracle):
    from jmodt_amd.ops.pointnet2.pointnet2_utils import farthest_point_sample
    assert not farthest_point_sample(T(np.ones((2, 300, 3), np.float32)), 20).cpu().numpy().any()
    g = load_golden("fps.npz")
    for tag in ("rand", "dup", "grid", "n1000", "n16384"):
        want = g[f"{tag}_idx"]
        assert np.array_equal(farthest_point_sample(T(g[f"{tag}_xyz"]), want.shape[1]).cpu().numpy(), want), tag


@pytest.mark.parametrize("variant", ["1", "2"])
def test_fps_pruned_variants_bit_exact(variant):
    """the spatially pruned kernels (tools/csrc/fps_pruned.hip; TOOLS build of the library only — the product
    library does not contain them; JM_FPS_PRUNE=1: wave clusters, =2: slot clusters) must give the same picks as the
    plain scan — golden cloud, duplicates (permanent ties), a grid (ties across clusters), identical points (every
    pair tied), 8192 and 16384 points"""
    import os
    import subprocess
    import sys
    from jmodt_amd.csrc import build as hip_build
    assert os.path.isfile(hip_build.TOOLS_LIB), "tools library missing: python -m jmodt_amd.csrc.build --tools"
    code = (
        "import sys, numpy as np, torch; sys.path.insert(0, %r)\n"
        "from jmodt_amd import synth, _lib\n"
        "from jmodt_amd.csrc import build as hip_build\n"
        "_lib.LIB_PATH = hip_build.TOOLS_LIB\n"
        "from jmodt_amd.ops.pointnet2.pointnet2_utils import farthest_point_sample\n"
        "from oracle import oracle as o\n"
        "g = np.load(%r)\n"
        "ok = np.array_equal(farthest_point_sample(torch.from_numpy(g['n16384_xyz']).cuda(), g['n16384_idx'].shape[1]).cpu().numpy(), g['n16384_idx'])\n"
        "cases = [synth.cloud(2, 4096, seed=9, dup_frac=0.2), synth.cloud(2, 16384, seed=10, dup_frac=0.3),\n"
        "         synth.cloud(1, 16384, seed=11, quantize=2.0 ** -1), synth.cloud(2, 8192, seed=12, dup_frac=0.1),\n"
        "         np.full((1, 16384, 3), 0.25, np.float32), synth.dense_cloud(1, 16384, 13)]\n"
        "for x in cases:\n"
        "    m = 600 if x.shape[1] > 4096 else 700\n"
        "    ok &= np.array_equal(farthest_point_sample(torch.from_numpy(x).cuda(), m).cpu().numpy(), o.furthest_point_sample(x, m))\n"
        "print('PRUNED_OK' if ok else 'PRUNED_MISMATCH')\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
         os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fps.npz"))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, JM_FPS_PRUNE=variant), capture_output=True,
                         text=True, timeout=600)
    assert "PRUNED_OK" in out.stdout, out.stdout + out.stderr


def test_fps_large_n_stream_path(oracle):
    """n > 16384 takes the streaming kernel (config 5: 65536 points)"""
    from jmodt_amd.ops.pointnet2.pointnet2_utils import farthest_point_sample
    xyz = synth.cloud(1, 20000, seed=5, dup_frac=0.05)
    assert np.array_equal(farthest_point_sample(T(xyz), 64).cpu().numpy(), oracle.furthest_point_sample(xyz, 64))


@pytest.mark.parametrize("B,N,m,kw", [
    (1, 20000, 64, dict(dup_frac=0.05)),       # 2 workgroups, the second partly filled
    (3, 65536, 300, {}),                       # config 5 cloud size: 4 workgroups per cloud
    (2, 40000, 200, dict(quantize=2.0 ** -2)), # heavy value ties across workgroups
    (1, 131072, 40, dict(dup_frac=0.3)),       # the largest supported cloud, 8 workgroups
    (9, 16385, 33, {}),                        # more clouds than XCDs, one point over the single-workgroup limit
])
def test_fps_cooperative_multi_workgroup_bit_exact(oracle, B, N, m, kw):
    """n > 16384: the cloud is split over co-operating workgroups (jm_furthest_point_sampling_ws)"""
    from jmodt_amd.ops.pointnet2.pointnet2_utils import farthest_point_sample
    xyz = synth.cloud(B, N, seed=N % 97, **kw)
    got = farthest_point_sample(T(xyz), m).cpu().numpy()
    assert np.array_equal(got, oracle.furthest_point_sample(xyz, m))


def test_fps_cooperative_all_equal_points():
    from jmodt_amd.ops.pointnet2.pointnet2_utils import farthest_point_sample
    assert not farthest_point_sample(T(np.full((2, 30000, 3), 0.5, np.float32)), 50).cpu().numpy().any()


def test_fps_full_size_properties():
    """B=8, 16384 -> 4096: indices in range, all distinct (distinct points), first is 0, and the
    min-distance of each new pick to the already picked set is non-increasing (FPS invariant)."""
    from jmodt_amd.ops.pointnet2.pointnet2_utils import farthest_point_sample
    xyz = synth.cloud(8, 16384, seed=1235)
    idx = farthest_point_sample(T(xyz), 4096).cpu().numpy()
    assert idx.shape == (8, 4096) and (idx[:, 0] == 0).all() and idx.min() >= 0 and idx.max() < 16384
    for b in range(8):
        assert np.unique(idx[b]).size == 4096
    p = torch.from_numpy(xyz[0][idx[0][:600]]).double()
    d = torch.cdist(p, p)
    gaps = [d[j, :j].min().item() for j in range(1, 600)]
    assert all(gaps[i] >= gaps[i + 1] - 1e-6 for i in range(len(gaps) - 1))


# ------------------------------------------------------------------ ball query / group / gather
@pytest.mark.parametrize("radius,nsample,dense,N,M", [
    (0.1, 16, False, 2048, 256), (0.5, 32, False, 2048, 256), (4.0, 64, False, 3000, 100),
    (0.4, 16, True, 1024, 128), (1.0, 32, True, 1024, 128), (2.0, 64, True, 512, 70), (1.0, 8, True, 37, 5),
    (0.2, 64, True, 512, 128),
])
def test_ball_query_bit_exact(oracle, radius, nsample, dense, N, M):
    from jmodt_amd.ops.pointnet2.pointnet2_utils import ball_query
    xyz = synth.dense_cloud(3, N, 5) if dense else synth.cloud(3, N, 5, dup_frac=0.1)
    new_xyz = np.ascontiguousarray(xyz[:, :M])
    got = ball_query(radius, nsample, T(xyz), T(new_xyz)).cpu().numpy()
    assert np.array_equal(got, oracle.ball_query(radius, nsample, xyz, new_xyz))


def test_ball_query_dual_and_golden(oracle):
    from jmodt_amd.ops.pointnet2.pointnet2_utils import ball_query, ball_query_dual
    g = load_golden("ball_query.npz")
    xyz, new = T(g["xyz"]), T(g["new_xyz"])
    i0, i1 = ball_query_dual(0.1, 16, 0.5, 32, xyz, new)
    assert np.array_equal(i0.cpu().numpy(), g["sparse_r0.1_ns16"]) and np.array_equal(i1.cpu().numpy(), g["sparse_r0.5_ns32"])
    assert np.array_equal(ball_query(4.0, 64, xyz, new).cpu().numpy(), g["sparse_r4.0_ns64"])
    dense, dnew = T(g["dense"]), T(g["dnew"])
    i0, i1 = ball_query_dual(0.4, 16, 1.0, 32, dense, dnew)
    assert np.array_equal(i0.cpu().numpy(), g["dense_r0.4_ns16"]) and np.array_equal(i1.cpu().numpy(), g["dense_r1.0_ns32"])
    assert np.array_equal(ball_query(2.0, 64, dense, dnew).cpu().numpy(), g["dense_r2.0_ns64"])


def test_ball_query_edge_cases():
    from jmodt_amd.ops.pointnet2.pointnet2_utils import ball_query
    xyz = np.zeros((1, 8, 3), np.float32)
    xyz[0, :, 0] = np.arange(8)
    centres = np.array([[[100, 0, 0], [0.0, 0, 0], [2.0, 0, 0], [3.5, 0, 0]]], np.float32)
    idx = ball_query(1.0, 4, T(xyz), T(centres)).cpu().numpy()
    assert idx[0].tolist() == [[0, 0, 0, 0], [0, 0, 0, 0], [2, 2, 2, 2], [3, 4, 3, 3]]
    assert ball_query(2.5, 3, T(xyz), T(centres[:, 2:3])).cpu().numpy()[0, 0].tolist() == [0, 1, 2]


def test_sa_level1_full_size_vs_oracle(oracle):
    """B=8 is BASELINE's batch; the oracle check runs on 2 frames x 512 centres to stay in seconds"""
    from jmodt_amd.ops.pointnet2 import pointnet2_utils as pu
    xyz = synth.cloud(8, 16384, seed=1235)
    txyz = T(xyz)
    fidx = pu.farthest_point_sample(txyz, 4096)
    new_xyz = pu.gather_operation(txyz.transpose(1, 2).contiguous(), fidx).transpose(1, 2).contiguous()
    assert torch.equal(new_xyz, torch.gather(txyz, 1, fidx.long().unsqueeze(-1).expand(-1, -1, 3)))
    i0, i1 = pu.ball_query_dual(0.1, 16, 0.5, 32, txyz, new_xyz)
    nx = new_xyz.cpu().numpy()
    for b in (0, 7):
        sub = np.ascontiguousarray(nx[b:b + 1, 1000:1512])
        assert np.array_equal(i0[b, 1000:1512].cpu().numpy(), oracle.ball_query(0.1, 16, xyz[b:b + 1], sub)[0])
        assert np.array_equal(i1[b, 1000:1512].cpu().numpy(), oracle.ball_query(0.5, 32, xyz[b:b + 1], sub)[0])
    # properties at full size: every centre is one of the points, so every list is non-empty, each
    # listed neighbour lies strictly inside the ball, and indices ascend until the back-fill starts
    for nb, r in ((i0, 0.1), (i1, 0.5)):
        pts = torch.gather(txyz.unsqueeze(1).expand(-1, 4096, -1, -1), 2,
                           nb.long().unsqueeze(-1).expand(-1, -1, -1, 3))
        d2 = ((pts.double() - new_xyz.double().unsqueeze(2)) ** 2).sum(-1)
        assert (d2 < r * r * (1 + 1e-6)).all()
        diff = nb[:, :, 1:] - nb[:, :, :-1]
        assert ((diff > 0) | (nb[:, :, 1:] == nb[:, :, :1])).all()
    # grouping == torch.gather at full size (bit-exact copy)
    feats = torch.randn(8, 96, 16384, device=DEV)
    grouped = pu.grouping_operation(feats, i1)
    want = torch.gather(feats.unsqueeze(2).expand(-1, -1, 4096, -1), 3, i1.long().unsqueeze(1).expand(-1, 96, -1, -1))
    assert torch.equal(grouped, want)


def test_group_gather_forward_backward(oracle):
    from jmodt_amd.ops.pointnet2 import pointnet2_utils as pu
    rng = np.random.default_rng(3)
    B, C, N, M, S = 2, 13, 200, 33, 5      # odd sizes: exercises the non-vectorised paths
    feats = rng.normal(size=(B, C, N)).astype(np.float32)
    gidx = rng.integers(0, N, (B, M, S)).astype(np.int32)
    idx = rng.integers(0, N, (B, M)).astype(np.int32)
    tf = T(feats).requires_grad_(True)
    out = pu.grouping_operation(tf, T(gidx))
    assert np.array_equal(out.detach().cpu().numpy(), oracle.grouping_operation(feats, gidx))
    g = rng.normal(size=out.shape).astype(np.float32)
    out.backward(T(g))
    assert np.allclose(tf.grad.cpu().numpy(), oracle.grouping_operation_grad(g, gidx, N), atol=1e-5)
    tf2 = T(feats).requires_grad_(True)
    out2 = pu.gather_operation(tf2, T(idx))
    assert np.array_equal(out2.detach().cpu().numpy(), oracle.gather_operation(feats, idx))
    g2 = rng.normal(size=out2.shape).astype(np.float32)
    out2.backward(T(g2))
    assert np.allclose(tf2.grad.cpu().numpy(), oracle.gather_operation_grad(g2, idx, N), atol=1e-5)
    # vectorised path (P*S % 4 == 0)
    gidx4 = rng.integers(0, N, (B, 32, 8)).astype(np.int32)
    assert np.array_equal(pu.grouping_operation(T(feats), T(gidx4)).cpu().numpy(), oracle.grouping_operation(feats, gidx4))


@pytest.mark.parametrize("shape", [(2, 19, 3000, 257, 32), (1, 8, 64, 50, 16), (3, 5, 500, 41, 7)])
def test_grouping_grad_on_back_filled_neighbour_lists(oracle, shape):
    """a4 backward on the lists ball_query writes (ball_query_gpu.cu:36-40: cnt distinct hits, then copies of the first): the kernel
    sums each run of equal consecutive indices inside a wave before its atomic; runs cross group, wave and workgroup boundaries here
    (odd nsample, whole groups of one index, lists without any repetition)"""
    from jmodt_amd.ops.pointnet2 import pointnet2_utils as pu
    B, C, N, M, S = shape
    rng = np.random.default_rng(sum(shape))
    gidx = np.empty((B, M, S), dtype=np.int32)
    for b in range(B):
        for m in range(M):
            cnt = int(rng.integers(1, S + 1)) if m % 5 else (1 if m % 2 else S)
            hits = np.sort(rng.choice(N, size=cnt, replace=False)).astype(np.int32)
            gidx[b, m, :cnt] = hits
            gidx[b, m, cnt:] = hits[0]
    gidx[:, 3:6] = gidx[:, 3:4, :1]            # three consecutive groups of ONE index: a run longer than a group
    feats = rng.normal(size=(B, C, N)).astype(np.float32)
    tf = T(feats).requires_grad_(True)
    out = pu.grouping_operation(tf, T(gidx))
    g = rng.normal(size=out.shape).astype(np.float32)
    out.backward(T(g))
    want = oracle.grouping_operation_grad(g, gidx, N)
    assert np.allclose(tf.grad.cpu().numpy(), want, rtol=1e-5, atol=1e-5 * max(1.0, float(np.abs(want).max())))


# ------------------------------------------------------------------ three_nn / interpolate
@pytest.mark.parametrize("n,m", [(300, 75), (1024, 256), (77, 2), (513, 131)])
def test_three_nn_interpolate(oracle, n, m):
    from jmodt_amd.ops.pointnet2 import pointnet2_utils as pu
    unknown = synth.cloud(2, n, 7, dup_frac=0.1)
    known = np.ascontiguousarray(unknown[:, :m])
    dist, idx = pu.three_nn(T(unknown), T(known))
    d2, oidx = oracle.three_nn(unknown, known)
    assert np.array_equal(idx.cpu().numpy(), oidx)
    assert np.array_equal(dist.cpu().numpy(), np.sqrt(d2))
    if m < 3:
        return
    rng = np.random.default_rng(2)
    feats = rng.normal(size=(2, 19, m)).astype(np.float32)
    w = 1.0 / (np.sqrt(d2) + 1e-8)
    w = (w / w.sum(2, keepdims=True)).astype(np.float32)
    tf = T(feats).requires_grad_(True)
    out = pu.three_interpolate(tf, idx, T(w))
    assert np.array_equal(out.detach().cpu().numpy(), oracle.three_interpolate(feats, oidx, w))
    g = rng.normal(size=out.shape).astype(np.float32)
    out.backward(T(g))
    assert np.allclose(tf.grad.cpu().numpy(), oracle.three_interpolate_grad(g, oidx, w, m), atol=1e-4)


def test_three_nn_golden():
    from jmodt_amd.ops.pointnet2 import pointnet2_utils as pu
    g = load_golden("three_nn_interp.npz")
    dist, idx = pu.three_nn(T(g["unknown"]), T(g["known"]))
    assert np.array_equal(idx.cpu().numpy(), g["idx"]) and np.array_equal(dist.cpu().numpy(), np.sqrt(g["dist2"]))
    assert np.array_equal(pu.three_interpolate(T(g["feats"]), idx, T(g["weight"])).cpu().numpy(), g["out"])


# ------------------------------------------------------------------ roipool3d
def test_roipool3d_reference_golden():
    """expected values produced by the reference's own roipool3d.cpp (see make_golden.py)"""
    from jmodt_amd.ops.roipool3d.roipool3d_utils import roipool3d_gpu
    g = load_golden("roipool3d_ref.npz")
    pooled, empty = roipool3d_gpu(T(g["pts"]), T(g["feat"]), T(g["boxes"]), 0.2, int(g["S"]))
    assert np.array_equal(pooled.cpu().numpy(), g["pooled"])
    assert np.array_equal(empty.cpu().numpy(), g["empty"])


@pytest.mark.parametrize("N,M,C,S", [(4096, 40, 9, 128), (16384, 128, 130, 512), (1000, 7, 2, 30), (300, 3, 0, 64)])
def test_roipool3d_vs_oracle(oracle, N, M, C, S):
    from jmodt_amd.ext import roipool3d_cuda
    B = 2
    pts = synth.dense_cloud(B, N, 31, extent=20.0)
    pts[..., 1] /= 10.0
    boxes = synth.proposals(pts, M, 32)
    boxes[0, 0, 0:3] = [500, 0, 500]
    boxes[1, 1, 3:6] = [60, 60, 60]
    boxes[0, 2, 3:6] = [0.8, 0.8, 1.2]
    feat = np.random.default_rng(33).normal(size=(B, N, C)).astype(np.float32)
    eb = oracle.enlarge_box3d(boxes, 0.2)
    want_p, want_e = oracle.roipool3d(pts, feat, eb, S)
    # reference contract (pre-zeroed outputs, empty rows untouched)
    pooled = torch.zeros((B, M, S, 3 + C), device=DEV)
    empty = torch.zeros((B, M), dtype=torch.int32, device=DEV)
    roipool3d_cuda.forward(T(pts), T(eb), T(feat), pooled, empty)
    assert np.array_equal(pooled.cpu().numpy(), want_p) and np.array_equal(empty.cpu().numpy(), want_e)
    # zero_empty contract (uninitialised outputs)
    pooled2 = torch.full((B, M, S, 3 + C), float("nan"), device=DEV)
    empty2 = torch.full((B, M), 77, dtype=torch.int32, device=DEV)
    roipool3d_cuda.forward(T(pts), T(eb), T(feat), pooled2, empty2, zero_empty=1)
    assert np.array_equal(pooled2.cpu().numpy(), want_p) and np.array_equal(empty2.cpu().numpy(), want_e)
    assert want_e.sum() >= 1 and (want_e == 0).sum() >= 1


# ------------------------------------------------------------------ iou3d / NMS
def test_overlap_iou_vs_oracle_and_golden(oracle):
    from jmodt_amd.ops.iou3d import iou3d_utils
    from jmodt_amd.ext import iou3d_cuda
    g = load_golden("iou3d_nms.npz")
    a, b = T(g["pair_a"]), T(g["pair_b"])
    ov = torch.empty((a.shape[0], b.shape[0]), device=DEV)
    iou3d_cuda.boxes_overlap_bev_gpu(a, b, ov)
    assert np.array_equal(ov.cpu().numpy(), g["overlap"])          # deterministic math: bit-exact
    assert np.array_equal(iou3d_utils.boxes_iou_bev(a, b).cpu().numpy(), g["iou"])
    sq = np.array([[0, 0, 2, 2, 0.0], [0, 0, 2, 2, np.pi / 4], [10, 10, 12, 12, 0.3]], np.float32)
    m = iou3d_utils.boxes_iou_bev(T(sq), T(sq)).cpu().numpy()
    assert abs(m[0, 0] - 1) < 1e-6 and m[0, 2] == 0
    assert abs(m[0, 1] - 8 * (np.sqrt(2) - 1) / (8 - 8 * (np.sqrt(2) - 1))) < 1e-5
    pts = synth.dense_cloud(1, 256, 3, extent=10.0)
    b3a, b3b = synth.proposals(pts, 70, 4)[0], synth.proposals(pts, 33, 5)[0]
    got = iou3d_utils.boxes_iou3d_gpu(T(b3a), T(b3b)).cpu().numpy()
    assert np.abs(got - oracle.boxes_iou3d(b3a, b3b)).max() < 1e-5


@pytest.mark.parametrize("n", [1, 63, 64, 65, 500, 1000, 6300])
@pytest.mark.parametrize("thresh", [0.1, 0.8, 0.85])
def test_nms_normal_bit_exact(oracle, n, thresh):
    from jmodt_amd.ops.iou3d.iou3d_utils import nms_normal_gpu
    boxes, scores = synth.bev_boxes(n, 100 + n)
    got = nms_normal_gpu(T(boxes), T(scores), thresh).cpu().numpy()
    assert got.dtype == np.int64
    assert np.array_equal(got, oracle.nms(boxes, scores, thresh, normal=True))


@pytest.mark.parametrize("n,thresh", [(100, 0.1), (400, 0.5), (64, 0.8), (2700, 0.8), (1, 0.5)])
def test_nms_rotated_bit_exact(oracle, n, thresh):
    from jmodt_amd.ops.iou3d.iou3d_utils import nms_gpu
    boxes, scores = synth.bev_boxes(n, 7 + n)
    got = nms_gpu(T(boxes), T(scores), thresh).cpu().numpy()
    assert np.array_equal(got, oracle.nms(boxes, scores, thresh, normal=False))


def test_nms_extension_level_api_and_golden(oracle):
    from jmodt_amd.ext import iou3d_cuda
    g = load_golden("iou3d_nms.npz")
    order = np.argsort(-g["scores"], kind="stable")
    keep = torch.zeros(1000, dtype=torch.int64)
    num = iou3d_cuda.nms_normal_gpu(T(g["boxes"][order]), keep, 0.8)
    assert np.array_equal(order[keep[:num].numpy()], g["normal_0.8"])
    order = np.argsort(-g["scores_rot"], kind="stable")
    keep = torch.zeros(300, dtype=torch.int64)
    num = iou3d_cuda.nms_gpu(T(g["boxes_rot"][order]), keep, 0.1)
    assert np.array_equal(order[keep[:num].numpy()], g["rot_0.1"])
    k, nk = iou3d_cuda.nms_device(torch.zeros((0, 5), device=DEV), 0.5, 1)
    assert int(nk.item()) == 0


def test_nms_idempotent_full_size():
    """property at RPN size: NMS of the kept set keeps everything (no kept pair exceeds thr)"""
    from jmodt_amd.ops.iou3d.iou3d_utils import nms_normal_gpu
    boxes, scores = synth.bev_boxes(6300, 9)
    tb, ts = T(boxes), T(scores)
    keep = nms_normal_gpu(tb, ts, 0.8)
    again = nms_normal_gpu(tb[keep], ts[keep], 0.8)
    assert again.numel() == keep.numel() and torch.equal(again, torch.arange(keep.numel(), device=DEV))
    assert torch.all(ts[keep][:-1] >= ts[keep][1:])


# ------------------------------------------------------------------ LI-Fusion gather
@pytest.mark.parametrize("channels_last", [False, True])
@pytest.mark.parametrize("C,H,W,N", [(8, 24, 80, 300), (64, 192, 640, 4096), (5, 7, 9, 33)])
def test_feature_gather_vs_oracle(oracle, channels_last, C, H, W, N):
    from jmodt_amd.ops.fusion import feature_gather
    rng = np.random.default_rng(5)
    fm = rng.normal(size=(2, C, H, W)).astype(np.float32)
    xy = rng.uniform(-1.1, 1.1, (2, N, 2)).astype(np.float32)
    xy[0, 0] = [-1, -1]; xy[0, 1] = [1, 1]; xy[0, 2] = [1.5, 0.2]
    tfm = T(fm)
    if channels_last:
        tfm = tfm.contiguous(memory_format=torch.channels_last)
    got = feature_gather(tfm, T(xy)).cpu().numpy()
    assert np.abs(got - oracle.feature_gather(fm, xy)).max() < 1e-5
    assert not got[0, :, 2].any()


def test_feature_gather_golden_and_grad():
    from jmodt_amd.ops.fusion import feature_gather
    import torch.nn.functional as F
    g = load_golden("feature_gather_ref.npz")
    assert np.abs(feature_gather(T(g["fmap"]), T(g["xy"])).cpu().numpy() - g["out"]).max() < 1e-5
    fm = T(g["fmap"]).requires_grad_(True)
    fm2 = T(g["fmap"]).requires_grad_(True)
    xy = T(g["xy"])
    go = torch.randn(2, 8, 300, device=DEV)
    feature_gather(fm, xy).backward(go)
    F.grid_sample(fm2, xy.unsqueeze(1), align_corners=True).squeeze(2).backward(go)
    assert (fm.grad - fm2.grad).abs().max().item() < 1e-4


# ------------------------------------------------------------------ affinity
def _heads_from_golden(g):
    from jmodt_amd.ops.affinity import make_affinity_mlp
    heads = []
    for name in ("link", "se"):
        h = make_affinity_mlp()
        h.load_state_dict({k[len(name) + 1:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(name + ".")})
        heads.append(h.to(DEV).eval())
    return heads


def test_affinity_reference_golden():
    """targets: the reference's layer builder + tracker.py:81-112 torch ops (make_golden.py)"""
    from jmodt_amd.ops.affinity import pairwise_affinity
    g = load_golden("affinity_ref.npz")
    link, se = _heads_from_golden(g)
    for tag in ("64x64", "3x5", "1x1"):
        A, s, e, raw = pairwise_affinity(T(g[f"{tag}_pf"]), T(g[f"{tag}_df"]), link, se, return_raw=True)
        assert np.abs(raw.cpu().numpy() - g[f"{tag}_raw"]).max() < 1e-4, tag
        assert np.abs(A.cpu().numpy() - g[f"{tag}_A"]).max() < 1e-4, tag
        assert np.abs(s.cpu().numpy() - g[f"{tag}_start"]).max() < 1e-4, tag
        assert np.abs(e.cpu().numpy() - g[f"{tag}_end"]).max() < 1e-4, tag


@pytest.mark.parametrize("P,D,C", [(7, 5, 64), (130, 67, 512), (256, 256, 512)])
def test_affinity_vs_oracle(oracle, P, D, C):
    from jmodt_amd.ops.affinity import make_affinity_mlp, mlp3_forward, pairwise_affinity
    lw, sw = synth.mlp_weights(C, C, C, 1), synth.mlp_weights(C, C, C, 2)

    def head(w):
        h = make_affinity_mlp(C, (C, C))
        with torch.no_grad():
            h[0].conv.weight.copy_(torch.from_numpy(w[0])[..., None]); h[0].conv.bias.copy_(torch.from_numpy(w[1]))
            h[2].conv.weight.copy_(torch.from_numpy(w[2])[..., None]); h[2].conv.bias.copy_(torch.from_numpy(w[3]))
            h[3].conv.weight.copy_(torch.from_numpy(w[4])[None, :, None]); h[3].conv.bias.fill_(float(w[5]))
        return h.to(DEV).eval()

    pf, df = synth.roi_features(P, C, 3), synth.roi_features(D, C, 4)
    A, s, e, raw = pairwise_affinity(T(pf), T(df), head(lw), head(sw), return_raw=True)
    if P * D <= 130 * 67:
        assert np.abs(raw.cpu().numpy() - oracle.link_scores(pf, df, lw)).max() < 1e-4
        oA, os_, oe = oracle.affinity(pf, df, lw, sw)
        assert np.abs(A.cpu().numpy() - oA).max() < 1e-4
        assert np.abs(s.cpu().numpy() - os_).max() < 1e-4 and np.abs(e.cpu().numpy() - oe).max() < 1e-4
    else:  # 256^2 (config 5): spot rows against the oracle + softmax properties at full size
        rows = [0, 100, 255]
        want = oracle.link_scores(pf[rows], df, lw)
        assert np.abs(raw[rows].cpu().numpy() - want).max() < 1e-4
        Ad = A.double()
        rd = torch.softmax(raw.double(), 1) + torch.softmax(raw.double(), 0)
        assert (Ad - rd / 2).abs().max().item() < 1e-5
    y = mlp3_forward(T(pf), head(sw))
    assert y.shape == (P,)


# ------------------------------------------------------------------ fused SA block (group + MLP + max-pool)
def _randomise_bn(module, seed):
    g = torch.Generator().manual_seed(seed)
    for m in module.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            with torch.no_grad():
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.2)
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.3)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)


@pytest.mark.parametrize("N,npoint,C,radii,nsamples,mlps,bn", [
    (2048, 512, 0, [0.5, 1.0], [16, 32], [[0, 16, 16, 32], [0, 32, 32, 64]], True),          # RPN level-1 shape
    (1024, 256, 96, [1.0, 2.0], [16, 32], [[96, 64, 64, 128], [96, 64, 96, 128]], True),     # RPN level-2 widths
    (512, 128, 128, [2.0], [64], [[128, 128, 128, 128]], True),                              # RCNN SA1 (config.py:137)
    (512, 128, 128, [2.0], [64], [[128, 128, 128, 256]], True),                              # last layer wider than a tile
    (700, 64, 5, [3.0], [32], [[5, 24, 40]], False),                                         # 2 layers, odd widths, no BN
    (600, 64, 300, [2.5], [32], [[300, 64, 96, 200]], True),                                 # 3 input chunks (303 channels)
    (600, 128, 253, [2.5], [16], [[253, 128, 128]], True),                                   # exactly 2 full chunks
    (512, 64, 40, [2.5], [64], [[40, 72]], True),                                            # single layer
    # ---- the wide kernel (sa_mlp_wide.hip): hidden widths > 128
    (1024, 256, 256, [1.0, 2.0], [16, 32], [[256, 128, 196, 256], [256, 128, 196, 256]], True),   # RPN SA3 (config.py:80)
    (256, 64, 512, [2.0, 4.0], [16, 32], [[512, 256, 256, 512], [512, 256, 384, 512]], True),     # RPN SA4 (config.py:81)
    (700, 64, 5, [3.0], [32], [[5, 200, 40]], False),                                        # 2 layers, odd widths, no BN
    (600, 63, 40, [2.5], [16], [[40, 160, 72]], True),                                       # a 32-row tile spanning two frames
    (300, 32, 0, [2.5], [32], [[0, 130, 300, 7]], True),                                     # no features, narrow output
])
def test_fused_sa_block_matches_unfused(N, npoint, C, radii, nsamples, mlps, bn):
    """fused kernel (eval, no-grad) vs the same module on the unfused path (HIP group ops + torch
    1x1 convs + BN + max), 1e-4 relative to the activation scale"""
    from jmodt_amd.ops.pointnet2.pointnet2_modules import PointnetSAModuleMSG
    torch.manual_seed(3)
    sa = PointnetSAModuleMSG(npoint=npoint, radii=radii, nsamples=nsamples, mlps=[list(m) for m in mlps], bn=bn)
    _randomise_bn(sa, 5)
    sa = sa.to(DEV).eval()
    xyz = T(synth.dense_cloud(2, N, 17, extent=6.0))
    feats = torch.randn(2, C, N, device=DEV) if C else None
    with torch.no_grad():
        sa.fuse = True
        nx1, f1, i1 = sa(xyz, feats)
        sa.fuse = False
        nx2, f2, i2 = sa(xyz, feats)
    assert torch.equal(nx1, nx2) and torch.equal(i1, i2) and f1.shape == f2.shape
    scale = f2.abs().max().item()
    assert scale > 0.1
    assert (f1 - f2).abs().max().item() <= 1e-4 * max(scale, 1.0), (f1 - f2).abs().max().item()


def test_fused_sa_block_repacks_after_weight_update():
    """the packed-weight cache follows in-place parameter updates (optimizer steps, load_state_dict)"""
    from jmodt_amd.ops.pointnet2.pointnet2_modules import PointnetSAModuleMSG
    torch.manual_seed(4)
    sa = PointnetSAModuleMSG(npoint=64, radii=[1.5], nsamples=[32], mlps=[[8, 32, 48]]).to(DEV).eval()
    xyz, feats = T(synth.dense_cloud(1, 400, 19, extent=5.0)), torch.randn(1, 8, 400, device=DEV)
    with torch.no_grad():
        a = sa(xyz, feats)[1].clone()
        for prm in sa.parameters():
            prm.mul_(1.5)
        b = sa(xyz, feats)[1]
        sa.fuse = False
        c = sa(xyz, feats)[1]
    assert (a - b).abs().max().item() > 1e-3
    assert (b - c).abs().max().item() <= 1e-4 * max(c.abs().max().item(), 1.0)


def test_xyz_only_scales_take_the_vector_pipe_kernel(oracle):
    """RPN level 1 (no input features, [3,16,16,32] x 16 and [3,32,32,64] x 32): jm_sa_mlp_supported says 3 = sa_xyz.hip; its
    output against a float64 evaluation of the same folded layers on the oracle's own neighbour lists"""
    import ctypes
    from jmodt_amd import _lib as L
    from jmodt_amd.ops.pointnet2 import fused
    from jmodt_amd.ops.pointnet2.pointnet2_modules import PointnetSAModuleMSG
    lib = L.load()
    for ns, w in ((16, [3, 16, 16, 32]), (32, [3, 32, 32, 64])):
        assert lib.jm_sa_mlp_supported(8, 16384, 4096, 0, ns, 0, 3, (ctypes.c_int * 4)(*w)) == 3
    assert lib.jm_sa_mlp_supported(8, 16384, 4096, 0, 16, 0, 3, (ctypes.c_int * 4)(3, 16, 24, 32)) == 1     # other widths: MFMA kernel
    assert lib.jm_sa_mlp_supported(2, 2048, 72, 0, 32, 0, 3, (ctypes.c_int * 4)(3, 32, 32, 64)) == 1         # centres not a multiple of 32
    torch.manual_seed(11)
    sa = PointnetSAModuleMSG(npoint=1024, radii=[0.1, 0.5], nsamples=[16, 32], mlps=[[0, 16, 16, 32], [0, 32, 32, 64]], bn=True)
    _randomise_bn(sa, 9)
    sa = sa.to(DEV).eval()
    xyz = synth.kitti_like_cloud(2, 8192, 23)
    with torch.no_grad():
        new_xyz, feats, _ = sa(T(xyz))
    new_np = new_xyz.cpu().numpy()
    off = 0
    for g, mlp, ns, r in zip(sa.groupers, sa.mlps, (16, 32), (0.1, 0.5)):
        idx = oracle.ball_query(r, ns, xyz, new_np).astype(np.int64)                  # (B, M, ns)
        rel = np.take_along_axis(xyz[:, None], idx[..., None], axis=2) - new_np[:, :, None, :]   # (B, M, ns, 3)
        h = torch.from_numpy(rel).double()
        for W, b in fused.fold_shared_mlp(mlp):
            h = torch.relu(h @ W.double().cpu().t() + b.double().cpu())
        want = h.max(dim=2)[0].permute(0, 2, 1)                                       # (B, C, M)
        got = feats[:, off:off + want.shape[1]].double().cpu()
        off += want.shape[1]
        assert (got - want).abs().max().item() <= 1e-4 * max(1.0, want.abs().max().item())
        assert want.abs().max().item() > 0.05


def test_fused_sa_block_is_used_and_falls_back():
    from jmodt_amd.ops.pointnet2 import fused
    from jmodt_amd.ops.pointnet2.pointnet2_modules import PointnetSAModuleMSG
    sa = PointnetSAModuleMSG(npoint=64, radii=[1.0], nsamples=[16], mlps=[[0, 16, 32]]).to(DEV)
    assert fused.can_fuse(sa.mlps[0], 64, 16, training=False)
    assert not fused.can_fuse(sa.mlps[0], 64, 16, training=True)        # batch statistics: not foldable
    assert not fused.can_fuse(sa.mlps[0], 64, 24, training=False)       # nsample not in {16,32,64}
    wide = PointnetSAModuleMSG(npoint=64, radii=[1.0], nsamples=[16], mlps=[[0, 196, 32]]).to(DEV)
    assert fused.can_fuse(wide.mlps[0], 64, 16, training=False)         # hidden width > 128: the wide kernel
    assert not fused.can_fuse(wide.mlps[0], 64, 64, training=False)     # ... which takes nsample 16 / 32 only
    huge = PointnetSAModuleMSG(npoint=64, radii=[1.0], nsamples=[16], mlps=[[0, 300, 32]]).to(DEV)
    assert not fused.can_fuse(huge.mlps[0], 64, 16, training=False)     # first layer wider than 256: un-fused operators


@pytest.mark.parametrize("B,N,C,mlp,bn", [(37, 32, 256, [256, 256, 256, 512], False),     # RCNN SA3 (config.py:139)
                                          (5, 32, 20, [20, 48, 96], True), (4, 16, 0, [0, 64, 33], True)])
def test_group_all_fused_matches_unfused(B, N, C, mlp, bn):
    """GroupAll + SharedMLP + max-pool (the RCNN's last SA level, pointnet2_utils.py:267-290) in the wide fused
    kernel vs the same module on the un-fused path (xyz NOT re-centred, one group of all N points)"""
    from jmodt_amd.ops.pointnet2 import fused
    from jmodt_amd.ops.pointnet2.pointnet2_modules import PointnetSAModule
    torch.manual_seed(8)
    sa = PointnetSAModule(mlp=list(mlp), npoint=None, radius=100.0, nsample=64, bn=bn)
    _randomise_bn(sa, 6)
    sa = sa.to(DEV).eval()
    xyz = T(synth.dense_cloud(B, N, 23, extent=3.0) - 1.5)
    feats = torch.randn(B, C, N, device=DEV) if C else None
    assert fused.can_fuse(sa.mlps[0], 1, N, False, B, N, group_all=True)
    with torch.no_grad():
        nx1, f1, i1 = sa(xyz, feats)
        sa.fuse = False
        nx2, f2, i2 = sa(xyz, feats)
    assert nx1 is None and nx2 is None and i1 is None and f1.shape == f2.shape == (B, mlp[-1], 1)
    scale = f2.abs().max().item()
    assert scale > 0.1 and (f1 - f2).abs().max().item() <= 1e-4 * max(scale, 1.0)
    # GroupAll by itself (a5): (B, 3 + C, 1, N) = [xyz^T | features], and features-only without use_xyz
    from jmodt_amd.ops.pointnet2.pointnet2_utils import GroupAll
    g = GroupAll(use_xyz=True)(xyz, None, feats)
    assert g.shape == (B, 3 + C, 1, N) and torch.equal(g[:, :3, 0], xyz.transpose(1, 2))
    if C:
        assert torch.equal(g[:, 3:, 0], feats) and torch.equal(GroupAll(use_xyz=False)(xyz, None, feats)[:, :, 0], feats)


def test_fp_module_forward_and_backward_vs_torch_restatement():
    """PointnetFPModule (pointnet2_modules.py:125-164) on the HIP three_nn / three_interpolate vs a pure-torch
    restatement (cdist top-3, gather) with the same MLP: forward 1e-4, feature gradients 1e-3"""
    from jmodt_amd.ops.pointnet2.pointnet2_modules import PointnetFPModule
    torch.manual_seed(2)
    fp = PointnetFPModule(mlp=[24 + 7, 32, 16]).to(DEV).eval()
    unknown = T(synth.cloud(2, 700, seed=31))
    known = T(synth.cloud(2, 90, seed=32))
    uf = torch.randn(2, 7, 700, device=DEV)
    kf = torch.randn(2, 24, 90, device=DEV, requires_grad=True)
    out = fp(unknown, known, uf, kf)
    out.sum().backward()
    g1 = kf.grad.clone()
    kf2 = kf.detach().clone().requires_grad_(True)
    d = torch.cdist(unknown.double(), known.double())
    dist, idx = torch.topk(d, 3, dim=2, largest=False)
    w = 1.0 / (dist.float() + 1e-8)
    w = w / w.sum(2, keepdim=True)
    gathered = torch.gather(kf2.unsqueeze(2).expand(-1, -1, 700, -1), 3, idx.unsqueeze(1).expand(-1, 24, -1, -1))
    interp = (gathered * w.unsqueeze(1)).sum(3)
    want = fp.mlp(torch.cat((interp, uf), 1).unsqueeze(-1)).squeeze(-1)
    want.sum().backward()
    assert out.shape == (2, 16, 700)
    assert (out - want).abs().max().item() < 1e-4
    assert (g1 - kf2.grad).abs().max().item() < 1e-3
    # broadcast branch (known is None)
    glob = torch.randn(2, 24, 1, device=DEV)
    o2 = fp(unknown, None, uf, glob)
    w2 = fp.mlp(torch.cat((glob.expand(-1, -1, 700), uf), 1).unsqueeze(-1)).squeeze(-1)
    assert (o2 - w2).abs().max().item() < 1e-5


# ------------------------------------------------------------------ tracker association cost (§8f row 1)
@pytest.mark.parametrize("P,D", [(20, 14), (64, 64), (1, 3), (130, 37)])
def test_association_cost_vs_oracle(oracle, P, D):
    from jmodt_amd.ops.association import association_cost, boxes_dist_gpu
    pts = synth.dense_cloud(1, 512, 3, extent=12.0)
    a, b = synth.proposals(pts, P, 4)[0], synth.proposals(pts, D, 5)[0]
    link = np.random.default_rng(0).random((P, D)).astype(np.float32)
    cost, iou, dist = association_cost(T(a), T(b), T(link), 0.5, 0.3, 0.2, return_parts=True)
    assert np.abs(dist.cpu().numpy() - oracle.boxes_dist(a, b)).max() < 1e-4
    assert np.abs(iou.cpu().numpy() - oracle.boxes_iou3d(a, b)).max() < 1e-5
    assert np.abs(cost.cpu().numpy() - oracle.association_cost(a, b, link, 0.5, 0.3, 0.2)).max() < 1e-4
    assert np.abs(boxes_dist_gpu(T(a), T(b)).cpu().numpy() - oracle.boxes_dist(a, b)).max() < 1e-4
    from jmodt_amd.ops.iou3d.iou3d_utils import boxes_iou3d_gpu
    assert torch.equal(boxes_iou3d_gpu(T(a), T(b)), iou)       # same kernel arithmetic as the iou3d op


def test_boxes_dist_and_canonical_vs_reference_geometry_golden():
    """device kernels vs vectors produced with the reference's own kitti_utils functions (geometry_ref.npz)"""
    from jmodt_amd.ops.association import boxes_dist_gpu
    from jmodt_amd.ops.roipool3d.roipool3d_utils import roipool3d_canonical_gpu
    g = load_golden("geometry_ref.npz")
    assert np.abs(boxes_dist_gpu(T(g["boxes_a"]), T(g["boxes_b"])).cpu().numpy() - g["boxes_dist"]).max() < 1e-4
    # each RoI pools exactly its own cloud here (one frame per RoI): the pooled points come back in index order,
    # so the canonical coordinates must equal the reference's transform of the same points
    rois, xyz = g["rois"], g["pooled_xyz"]
    M, S = xyz.shape[0], xyz.shape[1]
    big = rois.copy()
    big[:, 3:6] = 60.0                                    # every point of the frame is inside its (huge) RoI
    big[:, 1] = rois[:, 1] + 30.0
    feat = np.zeros((M, S, 1), np.float32)
    got, flag = roipool3d_canonical_gpu(T(xyz), T(feat), T(big[:, None, :]), 0.0, S)
    # centre subtraction uses the RoI's (x, y_bottom, z): rebuild the reference transform for the moved y
    want = g["canonical"].copy()
    want[..., 1] = xyz[..., 1] - big[:, None, 1]
    inside = np.abs(xyz[..., 0] - rois[:, None, 0]) <= 10.0
    inside &= np.abs(xyz[..., 2] - rois[:, None, 2]) <= 10.0
    assert inside.all() and not flag.cpu().numpy().any()
    assert np.abs(got.cpu().numpy()[:, 0, :, :3] - want).max() < 1e-4


# ------------------------------------------------------------------ roipool3d + canonical transformation (§8f row 3)
@pytest.mark.parametrize("B,N,M,C,S", [(2, 4096, 24, 6, 128), (2, 2048, 10, 5, 64), (3, 16384, 32, 130, 512)])
def test_roipool3d_canonical_vs_oracle_and_unfused(oracle, B, N, M, C, S):
    """the fused kernel == oracle restatement of proposal_target_layer.py:100-112 (1e-4), == our own
    roipool3d_gpu followed by the reference's torch ops; index behaviour (first S, cyclic padding, empty
    RoIs -> transform of the zero row) is inherited from roipool3d"""
    from jmodt_amd.ops.roipool3d.roipool3d_utils import roipool3d_canonical_gpu, roipool3d_gpu
    pts = synth.dense_cloud(B, N, 601, extent=14.0)
    pts[..., 1] = pts[..., 1] / 7.0
    boxes = synth.proposals(pts, M, 602)
    boxes[0, 0, 0:3] = [500, 0, 500]           # empty
    boxes[0, 1, 3:6] = [60, 60, 60]            # > S points
    boxes[-1, 2, 3:6] = [1.0, 0.8, 1.5]        # few points: cyclic padding
    feat = np.random.default_rng(603).normal(size=(B, N, C)).astype(np.float32)
    got, gflag = roipool3d_canonical_gpu(T(pts), T(feat), T(boxes), 0.2, S)
    want, wflag = oracle.roipool3d_canonical(pts, feat, boxes, 0.2, S)
    assert np.array_equal(gflag.cpu().numpy(), wflag)
    g = got.cpu().numpy()
    assert np.array_equal(g[..., 3:], want[..., 3:])                     # copied features: bit-exact
    assert np.abs(g[..., :3] - want[..., :3]).max() < 1e-4
    # the reference's own sequence of torch ops on top of the unfused kernel
    ref, _ = roipool3d_gpu(T(pts), T(feat), T(boxes), 0.2, S)
    tb = T(boxes)
    ref[:, :, :, 0:3] -= tb[:, :, 0:3].unsqueeze(2)
    cosa, sina = torch.cos(tb[..., 6]), torch.sin(tb[..., 6])
    R = torch.stack((torch.stack((cosa, -sina), -1), torch.stack((sina, cosa), -1)), -2)       # (B, M, 2, 2)
    ref[..., [0, 2]] = torch.matmul(ref[..., [0, 2]], R.transpose(-1, -2))
    assert (got - ref).abs().max().item() < 1e-4
    assert wflag[0, 0] == 1 and np.abs(g[0, 0, :, 3:]).max() == 0


# ------------------------------------------------------------------ batched NMS + RPN proposal selection (§8f row 2)
@pytest.mark.parametrize("normal", [1, 0])
def test_nms_batched_matches_single_problem_oracle(oracle, normal):
    """ragged problems (incl. empty, 1 box, one exactly max_boxes) through one jm_nms_batched call"""
    from jmodt_amd.ext import iou3d_cuda
    counts = [700, 0, 1, 64, 65, 333, 1000, 129]
    nmax = 1000
    boxes = np.zeros((len(counts), nmax, 5), np.float32)
    for p, c in enumerate(counts):
        b, s = synth.bev_boxes(max(c, 1), 50 + p)
        boxes[p, :c] = b[np.argsort(-s, kind="stable")][:c]
        boxes[p, c:] = np.nan    # padding must never be read as a box
    keep, num = iou3d_cuda.nms_batched_device(T(boxes), T(np.array(counts, np.int32)), 0.6, normal)
    keep, num = keep.cpu().numpy(), num.cpu().numpy()
    for p, c in enumerate(counts):
        want = oracle.nms_sorted(boxes[p, :c], 0.6, normal) if c else np.zeros(0, np.int64)
        assert num[p] == len(want), p
        assert np.array_equal(keep[p, :num[p]], want), p


@pytest.mark.parametrize("first_k", [1, 39, 89, 512, 2048])
@pytest.mark.parametrize("kind,thresh", [("clustered", 0.6), ("clustered", 0.85), ("piled", 0.3), ("sparse", 0.8)])
def test_nms_first_k_equals_truncated_mask_reduce(oracle, kind, thresh, first_k):
    """jm_nms_normal_first_k_batched (lazy greedy: each box against the boxes KEPT before it, stops at K) == the first K
    entries of jm_nms_batched's keep lists, ragged problems incl. empty / 1 / chunk-boundary sizes / 6300 boxes;
    `piled`: hundreds of near-copies per object (a handful of survivors out of thousands: the walk visits every box)"""
    from jmodt_amd.ext import iou3d_cuda
    counts = [6300, 0, 1, 64, 65, 512, 513, 2700, 1025]
    nmax = 6300
    boxes = np.zeros((len(counts), nmax, 5), np.float32)
    rng = np.random.default_rng(int(thresh * 100) + first_k)
    for p, c in enumerate(counts):
        if kind == "clustered":
            b, s = synth.bev_boxes(max(c, 1), 70 + p)
        elif kind == "piled":
            b, s = synth.bev_boxes(max(c, 1), 80 + p, jitter_clusters=False)
            b[:, :4] += rng.normal(0, 0.05, (len(b), 1)).astype(np.float32)
            b = b[rng.integers(0, max(1, len(b) // 40), len(b))] + rng.normal(0, 0.02, (len(b), 5)).astype(np.float32)
        else:
            b, s = synth.bev_boxes(max(c, 1), 90 + p, extent=2000.0)
        boxes[p, :c] = b[np.argsort(-s, kind="stable")][:c]
        boxes[p, c:] = np.nan
    tb, tc = T(boxes), T(np.array(counts, np.int32))
    full_keep, full_num = iou3d_cuda.nms_batched_device(tb, tc, thresh, 1)
    keep, num = iou3d_cuda.nms_normal_first_k_device(tb, tc, thresh, first_k)
    full_keep, full_num, keep, num = (a.cpu().numpy() for a in (full_keep, full_num, keep, num))
    for p, c in enumerate(counts):
        k = min(first_k, int(full_num[p]))
        assert num[p] == k, (p, num[p], k)
        assert np.array_equal(keep[p, :k], full_keep[p, :k]), p
    if first_k == 89:      # the oracle itself on the small problems
        for p in (2, 3, 4, 5, 6):
            want = oracle.nms_sorted(boxes[p, :counts[p]], thresh, 1)[:first_k]
            assert np.array_equal(keep[p, :num[p]], want)


@pytest.mark.parametrize("B,N,pre,post,thresh,nms_type,kw", [
    (4, 16384, 9000, 100, 0.8, "normal", dict(empty_far=(1,), empty_near=(2,))),   # TEST config (config.py:226-230)
    (2, 16384, 9000, 512, 0.85, "normal", {}),                                    # TRAIN config (config.py:201-205)
    (3, 2000, 9000, 100, 0.8, "rotate", dict(empty_far=(0,))),                     # fewer points than the budget
    (2, 600, 300, 40, 0.7, "normal", dict(empty_far=(1,))),                        # far band falls back to near[pre1:]
    (2, 300, 100, 500, 0.99, "normal", {}),                                        # post budget never reached
])
def test_proposal_select_distance_based(oracle, B, N, pre, post, thresh, nms_type, kw):
    from jmodt_amd.ops.proposal import distance_based_proposal
    scores, props = synth.rpn_output(B, N, seed=B * N + post, **kw)
    got_b, got_s = distance_based_proposal(T(scores), T(props), pre, post, thresh, nms_type)
    want_b, want_s = oracle.proposal_select(scores, props, pre, post, thresh, nms_type)
    assert got_b.shape == (B, post, 7) and got_s.shape == (B, post)
    assert np.array_equal(got_s.cpu().numpy(), want_s)
    assert np.array_equal(got_b.cpu().numpy(), want_b)
    assert (want_s != 0).any(axis=1).all()


@pytest.mark.parametrize("avg_by_bin", [True, False])
def test_decode_rpn_proposals_and_whole_proposal_layer(oracle, avg_by_bin):
    """RPN box decode vs the oracle restatement (1e-4; parity unpinned, see ops/proposal.py), then the whole
    ProposalLayer (decode + selection) on the device vs the oracle selection fed with the same decoded boxes"""
    from jmodt_amd.ops.proposal import decode_rpn_proposals, proposal_layer
    rng = np.random.default_rng(11)
    B, N = 3, 16384
    xyz = synth.cloud(B, N, seed=21)
    reg = rng.normal(0, 1.5, (B, N, 76)).astype(np.float32)
    scores = (rng.permutation(B * N).reshape(B, N).astype(np.float32) - B * N / 2) / np.float32(B * N / 8)
    dec = decode_rpn_proposals(T(xyz), T(reg), avg_by_bin=avg_by_bin)
    want = oracle.decode_rpn_proposals(xyz, reg, avg_by_bin=avg_by_bin)
    assert dec.shape == (B, N, 7)
    assert np.abs(dec.cpu().numpy() - want).max() < 1e-4
    boxes, sc = proposal_layer(T(scores), T(reg), T(xyz), avg_by_bin=avg_by_bin)
    wb, ws = oracle.proposal_select(scores, dec.cpu().numpy(), 9000, 100, 0.8, "normal")
    assert np.array_equal(sc.cpu().numpy(), ws) and np.array_equal(boxes.cpu().numpy(), wb)


def test_proposal_layer_train_budgets_vs_the_references_own_proposal_layer():
    """proposal_layer on the GPU with the TRAIN budgets / threshold against the reference's ProposalLayer output stored in
    train_ref.npz (tests/golden/make_golden_train.py)"""
    import os
    from jmodt_amd.ops.proposal import proposal_layer
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "train_ref.npz"))
    pre, post, thr = int(g["prop_params"][0]), int(g["prop_params"][1]), float(g["prop_params"][2])
    boxes, sc = proposal_layer(T(g["prop_cls"]), T(g["prop_reg"]), T(g["prop_xyz"]), pre_nms_top_n=pre, post_nms_top_n=post, nms_thresh=thr)
    assert np.abs(boxes.cpu().numpy() - g["prop_rois"]).max() < 1e-4 * max(1.0, np.abs(g["prop_rois"]).max())
    assert np.abs(sc.cpu().numpy() - g["prop_scores"]).max() < 1e-6


def test_proposal_select_score_based(oracle):
    from jmodt_amd.ops.proposal import score_based_proposal
    scores, props = synth.rpn_output(3, 4096, seed=77)
    got_b, got_s = score_based_proposal(T(scores), T(props), 1500, 100, 0.8)
    want_b, want_s = oracle.proposal_select(scores, props, 1500, 100, 0.8, distance_based=False)
    assert np.array_equal(got_s.cpu().numpy(), want_s) and np.array_equal(got_b.cpu().numpy(), want_b)


def test_fps_pyramid_side_stream_matches_sequential(oracle):
    """the FPS chain run ahead on a side stream hands the main stream the same indices / centres"""
    from jmodt_amd.ops.pointnet2.pyramid import FpsPyramid
    from jmodt_amd.ops.pointnet2.pointnet2_utils import ball_query
    xyz = synth.cloud(3, 4096, seed=41)
    t = T(xyz)
    for _ in range(3):   # repeated use recycles the side-stream buffers
        pyr = FpsPyramid(t, [1024, 256, 64])
        cur = xyz
        for k, m in enumerate((1024, 256, 64)):
            idx, new_xyz = pyr.level(k)
            nb = ball_query(0.8, 16, T(cur), new_xyz)           # a main-stream consumer
            want = oracle.furthest_point_sample(cur, m)
            assert np.array_equal(idx.cpu().numpy(), want)
            nxt = np.take_along_axis(cur, want[..., None].astype(np.int64), axis=1)
            assert np.array_equal(new_xyz.cpu().numpy(), nxt)
            assert np.array_equal(nb.cpu().numpy(), oracle.ball_query(0.8, 16, cur, nxt))
            cur = nxt


# ------------------------------------------------------------------ BASELINE configs[4] sizes (dense scene)
def test_dense_config_sizes_vs_oracle(oracle):
    """65536-point cloud: dual ball query, 3-NN, roipool3d + canonical transform for 256 RoIs — the index
    outputs bit-exact against the oracle at full size (one frame)"""
    from jmodt_amd.ops.pointnet2 import pointnet2_utils as pu
    from jmodt_amd.ops.roipool3d.roipool3d_utils import roipool3d_canonical_gpu
    N, m = 65536, 4096
    xyz = synth.cloud(1, N, seed=65)
    t = T(xyz)
    idx = pu.farthest_point_sample(t, m)
    new_xyz = pu.gather_operation(t.transpose(1, 2).contiguous(), idx).transpose(1, 2).contiguous()
    nx = new_xyz.cpu().numpy()
    i0, i1 = pu.ball_query_dual(0.1, 16, 0.5, 32, t, new_xyz)
    assert np.array_equal(i0.cpu().numpy(), oracle.ball_query(0.1, 16, xyz, nx))
    assert np.array_equal(i1.cpu().numpy(), oracle.ball_query(0.5, 32, xyz, nx))
    dist, nn = pu.three_nn(t, new_xyz)
    wd, wi = oracle.three_nn(xyz, nx)
    assert np.array_equal(nn.cpu().numpy(), wi) and np.abs(dist.cpu().numpy() - np.sqrt(wd)).max() < 1e-5
    boxes = synth.proposals(xyz, 256, 66)
    feat = np.random.default_rng(67).normal(size=(1, N, 16)).astype(np.float32)
    got, flag = roipool3d_canonical_gpu(t, T(feat), T(boxes), 0.2, 512)
    want, wflag = oracle.roipool3d_canonical(xyz, feat, boxes, 0.2, 512)
    assert np.array_equal(flag.cpu().numpy(), wflag)
    g = got.cpu().numpy()
    assert np.array_equal(g[..., 3:], want[..., 3:]) and np.abs(g[..., :3] - want[..., :3]).max() < 1e-4


@pytest.mark.parametrize("N,npoint,C,ns,mlp,extent", [(512, 128, 128, 64, [128, 128, 128, 128], 3.0),       # RCNN SA1
                                                      (128, 32, 128, 64, [128, 128, 128, 256], 3.0),       # RCNN SA2
                                                      (4096, 1024, 96, 32, [96, 64, 96, 128], 60.0),        # RPN SA2, KITTI-sized coordinates
                                                      (600, 64, 40, 16, [40, 32, 72, 48], 8.0),            # nsample 16: 8 centres per tile
                                                      (256, 64, 64, 16, [64, 32, 32, 32], 6.0),            # one 32-column block per layer: half the MFMA waves skip it
                                                      (256, 64, 61, 32, [61, 64, 80, 64], 6.0),            # hidden 80: 3 blocks, odd k-tile count
                                                      (128, 64, 64, 16, [64, 32, 48, 64], 6.0)])
@pytest.mark.parametrize("pm", [True, False])
def test_fused_sa_pre_projected_first_layer(N, npoint, C, ns, mlp, extent, pm):
    """pm: the two-layer point-major kernel (csrc/sa_mlp_pm.hip) or the k-major one (csrc/sa_mlp.hip).
    The first layer hoisted in front of the gather (u = W1 [xyz | f] + b1 per point, relu(u_j - W1x c_i) formed in the
    kernel: jm_sa_mlp_forward_pre) vs the same kernel with the first layer evaluated per (centre, sample) row vs the
    un-fused module; coordinates up to the KITTI range (the subtraction u_j - W1x c_i cancels W1x-scaled coordinates)"""
    from jmodt_amd.ops.pointnet2 import fused
    from jmodt_amd.ops.pointnet2.pointnet2_modules import PointnetSAModule
    torch.manual_seed(11)
    sa = PointnetSAModule(mlp=list(mlp), npoint=npoint, radius=extent / 6, nsample=ns, bn=True)
    _randomise_bn(sa, 7)
    sa = sa.to(DEV).eval()
    B = 3
    xyz = T(synth.dense_cloud(B, N, 29, extent=extent) + np.float32(extent))
    feats = torch.randn(B, C, N, device=DEV)
    assert fused._can_pre_project(sa.mlps[0], feats, torch.zeros(B, npoint, ns), npoint, ns)
    from jmodt_amd.profile import prof
    with torch.no_grad():
        fused.PM_KERNEL = pm
        prof.reset()
        prof.enabled = True
        try:
            nx, f_pre, _ = sa(xyz, feats)
            torch.cuda.synchronize()
            used = set(prof.records)
        finally:
            fused.PM_KERNEL = True
            prof.enabled = False
            prof.reset()
        assert any("sa_mlp_pm_forward" in k for k in used) == pm and any("sa_mlp_forward_pre" in k for k in used) == (not pm), used
        fused.PRE_PROJECT = False
        try:
            _, f_row, _ = sa(xyz, feats)
        finally:
            fused.PRE_PROJECT = True
        sa.fuse = False
        _, f_ref, _ = sa(xyz, feats)
    scale = max(f_ref.abs().max().item(), 1.0)
    assert f_ref.abs().max().item() > 0.1
    assert (f_pre - f_ref).abs().max().item() <= 1e-4 * scale, (f_pre - f_ref).abs().max().item()
    assert (f_row - f_ref).abs().max().item() <= 1e-4 * scale


# ------------------------------------------------------------------ hash-grid ball query (csrc/ball_query_grid.hip)
def _bq_both(oracle, xyz, new_xyz, radii, nss):
    """grid path (the ops' default for n >= 2048) and brute-force entry vs the oracle, single and dual"""
    import ctypes
    from jmodt_amd import _lib as L
    from jmodt_amd.ops.pointnet2 import pointnet2_utils as pu
    B, N, _ = xyz.shape
    M = new_xyz.shape[1]
    want = [oracle.ball_query(r, ns, xyz, new_xyz) for r, ns in zip(radii, nss)]
    tx, tn = T(xyz), T(new_xyz)
    assert L.load().jm_ball_query_workspace_bytes(B, N) > 0
    for r, ns, w in zip(radii, nss, want):
        assert np.array_equal(pu.ball_query(r, ns, tx, tn).cpu().numpy(), w), (r, ns)
        brute = torch.zeros((B, M, ns), dtype=torch.int32, device=DEV)
        L.check(L.load().jm_ball_query(B, N, M, float(r), ns, L.dev(tn, torch.float32, "c"), L.dev(tx, torch.float32, "x"),
                                       L.dev(brute, torch.int32, "i"), L.stream_ptr()), "bq")
        assert np.array_equal(brute.cpu().numpy(), w)
    if len(radii) == 2:
        i0, i1 = pu.ball_query_dual(radii[0], nss[0], radii[1], nss[1], tx, tn)
        assert np.array_equal(i0.cpu().numpy(), want[0]) and np.array_equal(i1.cpu().numpy(), want[1])
    return want


@pytest.mark.parametrize("kind", ["uniform", "kitti", "dense", "identical", "dup_heavy"])
def test_ball_query_grid_vs_oracle(oracle, kind):
    rng = np.random.default_rng(7)
    B, N, M = 2, 8192, 1024
    if kind == "uniform":
        xyz = synth.cloud(B, N, 31, dup_frac=0.1)
    elif kind == "kitti":
        xyz = synth.kitti_like_cloud(B, N, 32)
    elif kind == "dense":
        xyz = synth.dense_cloud(B, N, 33, extent=3.0)              # hundreds of points per ball: truncation at nsample
    elif kind == "identical":
        xyz = np.full((B, N, 3), 0.25, np.float32)                 # every point in one bucket, every pair a hit
        xyz[1, ::2] += 7.0
    else:
        xyz = synth.cloud(B, N, 34)
        xyz[:, N // 4:] = xyz[:, rng.integers(0, N // 4, N - N // 4)]   # 4 copies of everything on average
    pick = np.stack([rng.choice(N, M, replace=False) for _ in range(B)])
    new_xyz = np.take_along_axis(xyz, pick[..., None], axis=1).copy()
    new_xyz[:, -8:] += 300.0                                      # centres with no neighbour at all: rows stay at the caller's fill
    new_xyz[0, -9] = np.nan
    want = _bq_both(oracle, xyz, new_xyz, (0.1, 0.5), (16, 32))
    assert (want[1][:, -8:] == 0).all()
    if kind in ("dense", "identical"):
        assert (np.diff(want[1][:, :M - 9], axis=2) > 0).all()     # full rows of distinct ascending indices (truncated, no back-fill)
    _bq_both(oracle, xyz, new_xyz, (1.7,), (64,))


def test_ball_query_grid_full_size_and_far_coordinates(oracle):
    """B = 8 x 16384 -> 4096 (the level-1 call of the detector) on a slice, and a cloud 10^6 m from the origin (the padded
    reach then spans more than 64 cells: those centres scan the frame's whole sorted array)"""
    xyz = synth.kitti_like_cloud(8, 16384, 41)
    idx = oracle.furthest_point_sample(xyz[:2], 4096)
    new_xyz = np.take_along_axis(xyz[:2], idx[..., None].astype(np.int64), axis=1)
    from jmodt_amd.ops.pointnet2 import pointnet2_utils as pu
    full_new = np.concatenate([new_xyz] * 4)
    i0, i1 = pu.ball_query_dual(0.1, 16, 0.5, 32, T(xyz), T(full_new))
    for b in range(2):
        assert np.array_equal(i0[b].cpu().numpy(), oracle.ball_query(0.1, 16, xyz[b:b + 1], new_xyz[b:b + 1])[0])
        assert np.array_equal(i1[b].cpu().numpy(), oracle.ball_query(0.5, 32, xyz[b:b + 1], new_xyz[b:b + 1])[0])
    far = synth.cloud(1, 4096, 42) + np.float32(1.0e6)
    cen = far[:, :300].copy()
    _bq_both(oracle, far, cen, (2.0,), (16,))


def test_ball_query_prebuilt_grid_on_side_stream(oracle):
    """jm_ball_query_grid_build on a side stream, jm_ball_query_grid_query later on the main stream (the FpsPyramid use):
    same indices as the oracle; a grid reused for two different centre sets; a grid of the wrong cloud is refused"""
    import torch
    from jmodt_amd.ops.pointnet2 import pointnet2_utils as pu
    xyz = synth.kitti_like_cloud(2, 16384, 43)
    rng = np.random.default_rng(5)
    side = torch.cuda.Stream()
    txyz = T(xyz)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        grid = pu.BallQueryGrid(txyz, 0.5)
        ev = torch.cuda.Event(); ev.record()
    assert grid.ws is not None
    torch.cuda.current_stream().wait_event(ev)
    for seed in (0, 1):
        pick = np.stack([rng.choice(16384, 512, replace=False) for _ in range(2)])
        cen = np.take_along_axis(xyz, pick[..., None], axis=1).copy()
        i0, i1 = pu.ball_query_dual(0.1, 16, 0.5, 32, txyz, T(cen), grid=grid)
        assert np.array_equal(i0.cpu().numpy(), oracle.ball_query(0.1, 16, xyz, cen))
        assert np.array_equal(i1.cpu().numpy(), oracle.ball_query(0.5, 32, xyz, cen))
    assert not grid.matches(txyz[:, :4096])
    small = pu.BallQueryGrid(txyz[:, :1024].contiguous(), 0.5)      # below the grid's size range: no workspace, callers scan
    assert small.ws is None


# ------------------------------------------------------------------ hash-grid 3-NN (csrc/three_nn_grid.hip)
@pytest.mark.parametrize("kind", ["uniform", "kitti", "flat", "identical", "dup_heavy", "isolated"])
def test_three_nn_grid_vs_oracle(oracle, kind):
    """the grid walk (1024 <= m <= 16384) against the sequential scan: indices bit-exact incl. the earlier-index rule on
    equal distances, squared distances exact"""
    from jmodt_amd import _lib as L
    from jmodt_amd.ext import pointnet2_cuda
    from jmodt_amd.ops.pointnet2 import pointnet2_utils as pu
    rng = np.random.default_rng(3)
    B, N, M = 2, 6000, 2048
    if kind == "uniform":
        unknown = synth.cloud(B, N, 51)
    elif kind == "kitti":
        unknown = synth.kitti_like_cloud(B, N, 52)
    elif kind == "flat":
        unknown = synth.cloud(B, N, 53); unknown[..., 1] = 1.5                    # a plane: the box has no volume
    elif kind == "identical":
        unknown = np.full((B, N, 3), 2.0, np.float32); unknown[:, ::3] += 1.0     # two locations: every distance tied
    elif kind == "dup_heavy":
        unknown = synth.cloud(B, N, 54); unknown[:, N // 8:] = unknown[:, rng.integers(0, N // 8, N - N // 8)]
    else:
        unknown = synth.kitti_like_cloud(B, N, 55); unknown[:, :40] += 500.0      # far outliers: the full-scan list
        unknown[0, 41] = np.nan
    pick = np.stack([np.sort(rng.choice(N, M, replace=False)) for _ in range(B)])
    known = np.take_along_axis(unknown, pick[..., None], axis=1).copy()
    if kind == "isolated":
        known[:, :5] = unknown[:, :5]
    want_d2, want_idx = oracle.three_nn(unknown, known)
    ok = ~np.isnan(unknown).any(-1)
    d2, ii = _three_nn_grid(unknown, known)
    assert np.array_equal(ii[ok], want_idx[ok]) and np.array_equal(d2[ok], want_d2[ok])
    dist, idx = pu.three_nn(T(unknown), T(known))                  # the op (scan at this size) and the shim agree too
    assert np.array_equal(idx.cpu().numpy()[ok], want_idx[ok]) and np.array_equal(dist.cpu().numpy()[ok], np.sqrt(want_d2[ok]))
    d2s = torch.zeros((B, N, 3), device=DEV); iis = torch.zeros((B, N, 3), dtype=torch.int32, device=DEV)
    pointnet2_cuda.three_nn_wrapper(B, N, M, T(unknown), T(known), d2s, iis)
    assert np.array_equal(iis.cpu().numpy()[ok], want_idx[ok]) and np.array_equal(d2s.cpu().numpy()[ok], want_d2[ok])


def _three_nn_grid(unknown, known):
    """jm_three_nn_ws with the grid workspace: the walk, whatever the size policy says"""
    import ctypes
    from jmodt_amd import _lib as L
    B, N, _ = unknown.shape
    M = known.shape[1]
    lib = L.load()
    nbytes = lib.jm_three_nn_grid_workspace_bytes(B, N, M)
    assert nbytes > 0
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=DEV)
    d2 = torch.zeros((B, N, 3), device=DEV)
    ii = torch.zeros((B, N, 3), dtype=torch.int32, device=DEV)
    tu, tk = T(unknown), T(known)
    L.check(lib.jm_three_nn_ws(B, N, M, L.dev(tu, torch.float32, "u"), L.dev(tk, torch.float32, "k"), L.dev(d2, torch.float32, "d"),
                               L.dev(ii, torch.int32, "i"), ctypes.c_void_p(ws.data_ptr()), nbytes, L.stream_ptr()), "three_nn_ws")
    return d2.cpu().numpy(), ii.cpu().numpy()


def test_three_nn_grid_fp1_shape(oracle):
    """the last feature-propagation level of the detector: 16384 unknown points against their own 4096 FPS samples"""
    from jmodt_amd.ops.pointnet2 import pointnet2_utils as pu
    xyz = synth.kitti_like_cloud(2, 16384, 61)
    idx = oracle.furthest_point_sample(xyz, 4096)
    known = np.take_along_axis(xyz, idx[..., None].astype(np.int64), axis=1)
    want_d2, want_idx = oracle.three_nn(xyz, known)
    d2, got = _three_nn_grid(xyz, known)
    assert np.array_equal(got, want_idx) and np.array_equal(d2, want_d2)
    dist, got = pu.three_nn(T(xyz), T(known))
    assert np.array_equal(got.cpu().numpy(), want_idx)
    assert np.array_equal(dist.cpu().numpy(), np.sqrt(want_d2))


def test_three_nn_policy_takes_the_grid_for_the_dense_shape(oracle):
    """65536 unknown x 4096 known (BASELINE configs[4]): the op's default path is the grid walk; a slice against the oracle"""
    from jmodt_amd import _lib as L
    from jmodt_amd.ops.pointnet2 import pointnet2_utils as pu
    assert L.load().jm_three_nn_workspace_bytes(1, 65536, 4096) > 0 and L.load().jm_three_nn_workspace_bytes(8, 16384, 4096) == 0
    xyz = synth.kitti_like_cloud(1, 65536, 71)
    known = xyz[:, ::16].copy()
    dist, got = pu.three_nn(T(xyz), T(known))
    want_d2, want_idx = oracle.three_nn(xyz[:, :3000], known)
    assert np.array_equal(got.cpu().numpy()[:, :3000], want_idx) and np.array_equal(dist.cpu().numpy()[:, :3000], np.sqrt(want_d2))


@pytest.mark.parametrize("B,N", [(1, 1), (3, 5), (2, 1000), (2, 1024), (2, 1025), (8, 4096), (3, 5000), (2, 8192), (8, 16384), (2, 16383)])
def test_argsort_desc_stable_equals_torch_sort(B, N):
    """csrc/sort.hip (LDS radix argsort, one workgroup per row) == torch.sort(descending=True, stable=True)[1] bit for bit: random
    scores, heavily tied scores (8 distinct values), +-0.0, +-inf and NaN in one row"""
    from jmodt_amd.ops.proposal import argsort_desc_stable
    g = torch.Generator().manual_seed(N * 7 + B)
    rows = [torch.randn(B, N, generator=g), torch.randint(0, 8, (B, N), generator=g).float() - 3.5,
            torch.randn(B, N, generator=g).round(decimals=1)]
    special = torch.randn(B, N, generator=g)
    vals = torch.tensor([0.0, -0.0, float("inf"), float("-inf"), float("nan"), -float("nan"), 1e-45, -1e-45])
    special[:, torch.randperm(N, generator=g)[:min(N, 64)]] = vals[torch.randint(0, 8, (min(N, 64),), generator=g)]
    rows.append(special)
    for sc in rows:
        sc = sc.to(DEV).contiguous()
        got = argsort_desc_stable(sc)
        want = torch.sort(sc, dim=1, descending=True, stable=True)[1]
        assert got.dtype == torch.int64 and torch.equal(got, want)
