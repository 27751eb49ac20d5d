"""GPU tier (-m gpu): VALUES through the Level-1 extension-module shims (jmodt_amd/ext/*.py = the reference's pybind
tables pointnet2_api.cpp:10-24, roipool3d.cpp:198-203), called with the reference's POSITIONAL argument order exactly as
jmodt/ops/pointnet2/pointnet2_utils.py:27,59,67,97,130,148,176,192,220 and roipool3d_utils.py:26 call them, against the
C oracle.  Every shape has pairwise different b, c, n, m, nsample so that a swapped (n, m) / (c, n) pair in a shim cannot
produce the right answer by accident."""
import numpy as np
import pytest
import torch

from jmodt_amd import synth
from jmodt_amd.ext import pointnet2_cuda, roipool3d_cuda

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


B, C, N, M, NS = 3, 5, 700, 96, 16


@pytest.fixture(scope="module")
def scene():
    xyz = synth.dense_cloud(B, N, 11, extent=3.0)
    xyz[:, N - 40:] = xyz[:, :40]                                   # duplicates
    rng = np.random.default_rng(12)
    feats = rng.normal(size=(B, C, N)).astype(np.float32)
    return dict(xyz=xyz, feats=feats, rng=rng)


def test_fps_gather_ball_group_shims(scene, oracle):
    xyz, feats = scene["xyz"], scene["feats"]
    # pointnet2_utils.py:25-27: temp = 1e10, idx IntTensor(B, npoint); wrapper(B, N, npoint, xyz, temp, output)
    temp = torch.full((B, N), 1e10, device=DEV)
    idx = torch.zeros((B, M), dtype=torch.int32, device=DEV)
    assert pointnet2_cuda.farthest_point_sampling_wrapper(B, N, M, T(xyz), temp, idx) == 1
    want_idx, want_temp = oracle.furthest_point_sample(xyz, M, return_temp=True)
    assert np.array_equal(idx.cpu().numpy(), want_idx) and np.array_equal(temp.cpu().numpy(), want_temp)
    # :56-59 gather_points_wrapper(B, C, N, npoint, features, idx, output)
    out = torch.zeros((B, C, M), device=DEV)
    assert pointnet2_cuda.gather_points_wrapper(B, C, N, M, T(feats), idx, out) == 1
    assert np.array_equal(out.cpu().numpy(), oracle.gather_operation(feats, want_idx))
    # :66-68 gather_points_grad_wrapper(B, C, N, npoint, grad_out, idx, grad_features [zeros])
    g = scene["rng"].normal(size=(B, C, M)).astype(np.float32)
    gf = torch.zeros((B, C, N), device=DEV)
    assert pointnet2_cuda.gather_points_grad_wrapper(B, C, N, M, T(g), idx, gf) == 1
    assert np.allclose(gf.cpu().numpy(), oracle.gather_operation_grad(g, want_idx, N), atol=1e-5)
    # :216-220 ball_query_wrapper(B, N, npoint, radius, nsample, new_xyz, xyz, idx [zeros])
    new_xyz = np.take_along_axis(xyz, want_idx[..., None].astype(np.int64), axis=1)
    nb = torch.zeros((B, M, NS), dtype=torch.int32, device=DEV)
    assert pointnet2_cuda.ball_query_wrapper(B, N, M, 0.45, NS, T(new_xyz), T(xyz), nb) == 1
    want_nb = oracle.ball_query(0.45, NS, xyz, new_xyz)
    assert np.array_equal(nb.cpu().numpy(), want_nb)
    assert len(np.unique(want_nb)) > M                               # real neighbourhoods, not only back-fill
    # :172-176 group_points_wrapper(B, C, N, nfeatures, nsample, features, idx, output)
    grouped = torch.zeros((B, C, M, NS), device=DEV)
    assert pointnet2_cuda.group_points_wrapper(B, C, N, M, NS, T(feats), nb, grouped) == 1
    assert np.array_equal(grouped.cpu().numpy(), oracle.grouping_operation(feats, want_nb))
    # :188-192 group_points_grad_wrapper(B, C, N, npoint, nsample, grad_out, idx, grad_features [zeros])
    gg = scene["rng"].normal(size=(B, C, M, NS)).astype(np.float32)
    gfeat = torch.zeros((B, C, N), device=DEV)
    assert pointnet2_cuda.group_points_grad_wrapper(B, C, N, M, NS, T(gg), nb, gfeat) == 1
    assert np.allclose(gfeat.cpu().numpy(), oracle.grouping_operation_grad(gg, want_nb, N), atol=1e-4)


def test_three_nn_and_interpolate_shims(scene, oracle):
    xyz, rng = scene["xyz"], scene["rng"]
    n, m, c = N, 57, 7                                               # unknown (B, n, 3), known (B, m, 3), features (B, c, m)
    known = np.ascontiguousarray(xyz[:, ::N // m][:, :m])
    assert known.shape[1] == m
    # pointnet2_utils.py:93-97 three_nn_wrapper(B, N, m, unknown, known, dist2, idx)
    dist2 = torch.zeros((B, n, 3), device=DEV)
    idx = torch.zeros((B, n, 3), dtype=torch.int32, device=DEV)
    assert pointnet2_cuda.three_nn_wrapper(B, n, m, T(xyz), T(known), dist2, idx) is None
    want_d2, want_idx = oracle.three_nn(xyz, known)
    assert np.array_equal(idx.cpu().numpy(), want_idx) and np.array_equal(dist2.cpu().numpy(), want_d2)
    w = rng.random((B, n, 3)).astype(np.float32)
    w /= w.sum(2, keepdims=True)
    feats = rng.normal(size=(B, c, m)).astype(np.float32)
    # :126-130 three_interpolate_wrapper(B, c, m, n, features, idx, weight, output)
    out = torch.zeros((B, c, n), device=DEV)
    assert pointnet2_cuda.three_interpolate_wrapper(B, c, m, n, T(feats), idx, T(w), out) is None
    assert np.allclose(out.cpu().numpy(), oracle.three_interpolate(feats, want_idx, w), atol=1e-6)
    # :144-148 three_interpolate_grad_wrapper(B, c, n, m, grad_out, idx, weight, grad_features [zeros])  -- (n, m) swap places
    g = rng.normal(size=(B, c, n)).astype(np.float32)
    gf = torch.zeros((B, c, m), device=DEV)
    assert pointnet2_cuda.three_interpolate_grad_wrapper(B, c, n, m, T(g), idx, T(w), gf) is None
    assert np.allclose(gf.cpu().numpy(), oracle.three_interpolate_grad(g, want_idx, w, m), atol=1e-4)


@pytest.mark.parametrize("fn", ["forward", "forward_slow"])
def test_roipool3d_forward_and_forward_slow_shims(fn, oracle):
    """roipool3d_utils.py:20-26: boxes enlarged by the caller; forward(xyz, boxes3d, pts_feature, pooled [zeros],
    empty_flag [zeros]); the reference's forward_slow (roipool3d.cpp:200) computes the same result"""
    b, n, m, c, s = 2, 3000, 9, 6, 40
    xyz = synth.cloud(b, n, 21)
    xyz[:, :, 0] *= 0.2; xyz[:, :, 2] = xyz[:, :, 2] * 0.2 + 5.0      # dense enough for boxes to hold > s points
    boxes = oracle.enlarge_box3d(synth.proposals(xyz, m, 22), 0.2)
    boxes[0, 0, :3] = 500.0                                           # an empty box
    feats = np.random.default_rng(23).normal(size=(b, n, c)).astype(np.float32)
    pooled = torch.zeros((b, m, s, 3 + c), device=DEV)
    empty = torch.zeros((b, m), dtype=torch.int32, device=DEV)
    assert getattr(roipool3d_cuda, fn)(T(xyz), T(boxes), T(feats), pooled, empty) == 1
    wp, we = oracle.roipool3d(xyz, feats, boxes, s)
    assert np.array_equal(pooled.cpu().numpy(), wp) and np.array_equal(empty.cpu().numpy(), we)
    assert we[0, 0] == 1 and we.sum() < b * m


# ------------------------------------------------------------------ the reference's PYTHON layer (tests/golden/glue_ref.npz)
def test_operator_and_module_layer_against_the_references_python_layer():
    """jmodt_amd.ops.{pointnet2, iou3d, roipool3d} — same function / class names and call forms as the reference's files —
    against glue_ref.npz: the outputs of the REFERENCE's own pointnet2_utils.py / pointnet2_modules.py / iou3d_utils.py /
    roipool3d_utils.py executed in the authoring container over the CPU oracle's extension entry points
    (tests/golden/make_golden_glue.py).  Pins argument orders, zero-fill / back-fill conventions, QueryAndGroup's
    concatenation order, GroupAll, three_nn's sqrt, the interpolation weights, the eval-mode module composition (also on the
    fused kernels), the 3-D IoU arithmetic, the NMS score order and the box enlargement of roipool3d_gpu to the reference's code."""
    import os
    from jmodt_amd.ops.iou3d import iou3d_utils
    from jmodt_amd.ops.pointnet2 import pointnet2_modules as mods
    from jmodt_amd.ops.pointnet2 import pointnet2_utils as pu
    from jmodt_amd.ops.roipool3d import roipool3d_utils
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "glue_ref.npz"))
    G = lambda k: torch.from_numpy(g[k]).to(DEV)      # noqa: E731

    def close(got, key, tol=1e-4):
        want = g[key]
        got = got.detach().cpu().numpy()
        assert got.shape == want.shape, (key, got.shape, want.shape)
        assert np.abs(got - want).max() <= tol * max(1.0, np.abs(want).max()), (key, np.abs(got - want).max())

    # ---- operator layer
    xyz, feats = G("op_xyz"), G("op_feats")
    fps = pu.farthest_point_sample(xyz, 96)
    assert np.array_equal(fps.cpu().numpy(), g["op_fps"])
    new_xyz = pu.gather_operation(xyz.transpose(1, 2).contiguous(), fps).transpose(1, 2).contiguous()
    assert np.array_equal(new_xyz.cpu().numpy(), g["op_new_xyz"])
    bq = pu.ball_query(0.9, 16, xyz, new_xyz)
    assert np.array_equal(bq.cpu().numpy(), g["op_ball"])
    f = feats.clone().requires_grad_(True)
    grouped = pu.grouping_operation(f, bq)
    assert np.array_equal(grouped.detach().cpu().numpy(), g["op_grouped"])
    (grouped * G("op_group_w")).sum().backward()
    close(f.grad, "op_group_grad", 1e-5)
    assert np.array_equal(pu.QueryAndGroup(0.9, 16, use_xyz=True)(xyz, new_xyz, feats).cpu().numpy(), g["op_qg"])
    assert np.array_equal(pu.QueryAndGroup(0.9, 16, use_xyz=True)(xyz, new_xyz, None).cpu().numpy(), g["op_qg_nofeat"])
    assert np.array_equal(pu.QueryAndGroup(0.9, 16, use_xyz=False)(xyz, new_xyz, feats).cpu().numpy(), g["op_qg_noxyz"])
    assert np.array_equal(pu.GroupAll(use_xyz=True)(xyz, None, feats).cpu().numpy(), g["op_group_all"])
    dist, nn_idx = pu.three_nn(xyz, new_xyz)
    assert np.array_equal(nn_idx.cpu().numpy(), g["op_nn_idx"])
    close(dist, "op_nn_dist", 1e-6)
    w = 1.0 / (dist + 1e-8)
    w = w / w.sum(dim=2, keepdim=True)
    kf = G("op_known_feats").requires_grad_(True)
    interp = pu.three_interpolate(kf, nn_idx, w)
    close(interp, "op_interp", 1e-5)
    (interp * G("op_interp_w")).sum().backward()
    close(kf.grad, "op_interp_grad", 1e-5)
    gs = feats.clone().requires_grad_(True)
    gathered = pu.gather_operation(gs, fps)
    assert np.array_equal(gathered.detach().cpu().numpy(), g["op_gathered"])
    (gathered * G("op_gather_w")).sum().backward()
    close(gs.grad, "op_gather_grad", 1e-5)

    # ---- module layer: the reference's state dicts load as they are (same parameter names)
    def load(module, prefix):
        sd = {k[len(prefix) + 1:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(prefix + ".")}
        missing, unexpected = module.load_state_dict(sd, strict=False)
        assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing), (missing, unexpected)
        return module.to(DEV).eval()
    sa = load(mods.PointnetSAModuleMSG(npoint=64, radii=[0.8, 1.6], nsamples=[16, 32], mlps=[[7, 16, 16, 32], [7, 16, 24, 48]],
                                       use_xyz=True, bn=True), "sa")
    sa_all = load(mods.PointnetSAModule(mlp=[32 + 48, 64, 96], use_xyz=True, bn=True), "sa_all")
    fp = load(mods.PointnetFPModule(mlp=[80 + 7, 40, 24], bn=True), "fp")
    for fuse in (True, False):                          # the fused kernels and the operator route
        for m in (sa, sa_all):
            m.fuse = fuse
        with torch.no_grad():
            sx, sf = sa(xyz, feats)[:2]
            af = sa_all(sx, sf)[1]
            ff = fp(xyz, sx, feats, sf)
        assert np.array_equal(sx.cpu().numpy(), g["mod_sa_xyz"])
        close(sf, "mod_sa_feat"); close(af, "mod_all_feat"); close(ff, "mod_fp_feat")

    # ---- iou3d_utils / roipool3d_utils
    a3, b3 = G("iou_a"), G("iou_b")
    close(iou3d_utils.boxes_iou_bev(iou3d_utils.boxes3d_to_bev_torch(a3), iou3d_utils.boxes3d_to_bev_torch(b3)), "iou_bev", 1e-5)
    close(iou3d_utils.boxes_iou3d_gpu(a3, b3), "iou_3d", 1e-5)
    assert np.array_equal(iou3d_utils.nms_gpu(G("nms_boxes"), G("nms_scores"), 0.3).cpu().numpy(), g["nms_keep_rot"])
    assert np.array_equal(iou3d_utils.nms_normal_gpu(G("nms_boxes"), G("nms_scores"), 0.5).cpu().numpy(), g["nms_keep_normal"])
    # tracker cost-matrix terms (jmodt/tracking/data_association.py:10-28 boxes_dist_gpu, :42-44)
    from jmodt_amd.ops.association import association_cost, boxes_dist_gpu
    close(boxes_dist_gpu(G("assoc_pred"), G("assoc_det")), "assoc_dist", 1e-5)
    link = torch.linspace(0, 1, g["assoc_iou"].size, device=DEV).view(*g["assoc_iou"].shape)
    cost, iou, dist = association_cost(G("assoc_pred"), G("assoc_det"), link, 0.5, 0.3, 0.2, return_parts=True)
    close(iou, "assoc_iou", 1e-5); close(dist, "assoc_dist", 1e-5)
    assert (cost.cpu().numpy() - (link.cpu().numpy() * 0.5 + g["assoc_iou"] * 0.3 + g["assoc_dist"] * 0.2)).__abs__().max() < 1e-5
    pooled, empty = roipool3d_utils.roipool3d_gpu(G("roi_pts"), G("roi_feat"), G("roi_boxes"), 0.2, sampled_pt_num=64)
    assert np.array_equal(pooled.cpu().numpy(), g["roi_pooled"]) and np.array_equal(empty.cpu().numpy(), g["roi_empty"])
