"""CPU tier: the oracle against the committed golden vectors (tests/golden/, see make_golden.py),
the host-side API surface, and the C-ABI library's exported symbols."""
import ctypes
import os
import re

import numpy as np
import pytest

from tests.conftest import ROOT, load_golden


# ------------------------------------------------------------------ oracle vs golden
def test_fps_golden(oracle):
    g = load_golden("fps.npz")
    for tag in ("rand", "dup", "grid", "n1000", "n16384"):
        want = g[f"{tag}_idx"]
        assert np.array_equal(oracle.furthest_point_sample(g[f"{tag}_xyz"], want.shape[1]), want), tag


def test_ball_query_golden(oracle):
    g = load_golden("ball_query.npz")
    for key in g.files:
        m = re.match(r"(sparse|dense)_r([\d.]+)_ns(\d+)", key)
        if not m:
            continue
        xyz, new = (g["xyz"], g["new_xyz"]) if m.group(1) == "sparse" else (g["dense"], g["dnew"])
        assert np.array_equal(oracle.ball_query(float(m.group(2)), int(m.group(3)), xyz, new), g[key]), key


def test_three_nn_interp_golden(oracle):
    g = load_golden("three_nn_interp.npz")
    d2, idx = oracle.three_nn(g["unknown"], g["known"])
    assert np.array_equal(d2, g["dist2"]) and np.array_equal(idx, g["idx"])
    assert np.array_equal(oracle.three_interpolate(g["feats"], g["idx"], g["weight"]), g["out"])


def test_iou3d_nms_golden(oracle):
    g = load_golden("iou3d_nms.npz")
    for thr in (0.1, 0.8, 0.85):
        assert np.array_equal(oracle.nms(g["boxes"], g["scores"], thr, normal=True), g[f"normal_{thr}"])
    for thr in (0.1, 0.5):
        assert np.array_equal(oracle.nms(g["boxes_rot"], g["scores_rot"], thr, normal=False), g[f"rot_{thr}"])
    assert np.array_equal(oracle.boxes_overlap_bev(g["pair_a"], g["pair_b"]), g["overlap"])
    assert np.array_equal(oracle.boxes_iou_bev(g["pair_a"], g["pair_b"]), g["iou"])


def test_roipool3d_reference_golden(oracle):
    """expected outputs were produced by the reference's own roipool3d.cpp CPU functions"""
    g = load_golden("roipool3d_ref.npz")
    assert str(g["source"]) == "reference"
    assert np.array_equal(oracle.enlarge_box3d(g["boxes"], 0.2), g["enlarged"])
    pooled, empty = oracle.roipool3d(g["pts"], g["feat"], g["enlarged"], int(g["S"]))
    assert np.array_equal(pooled, g["pooled"]) and np.array_equal(empty, g["empty"])
    for b in range(2):
        assert np.array_equal(oracle.pts_in_boxes3d(g["pts"][b], g["enlarged"][b]).astype(np.uint8), g["flags"][b])
    assert g["empty"][0, 0] == 1 and g["flags"][0, 1].sum() > int(g["S"]) and 0 < g["flags"][1, 2].sum() < int(g["S"])


def test_kitti_utils_reference_golden(oracle):
    g = load_golden("kitti_utils_ref.npz")
    assert np.array_equal(oracle.boxes3d_to_bev(g["boxes3d"]), g["bev"])
    assert np.array_equal(oracle.enlarge_box3d(g["boxes3d"], 0.2), g["enlarged"])


def _golden_heads(g):
    def w(name):
        return (g[f"{name}.0.conv.weight"][..., 0], g[f"{name}.0.conv.bias"], g[f"{name}.2.conv.weight"][..., 0],
                g[f"{name}.2.conv.bias"], g[f"{name}.3.conv.weight"].reshape(-1), g[f"{name}.3.conv.bias"])
    return w("link"), w("se")


def test_affinity_reference_golden(oracle):
    """expected outputs: reference layer builder + the torch ops of tracker.py:81-112"""
    g = load_golden("affinity_ref.npz")
    link, se = _golden_heads(g)
    for tag in ("64x64", "3x5", "1x1"):
        pf, df = g[f"{tag}_pf"], g[f"{tag}_df"]
        assert np.abs(oracle.link_scores(pf, df, link) - g[f"{tag}_raw"]).max() < 1e-4
        A, s, e = oracle.affinity(pf, df, link, se)
        assert np.abs(A - g[f"{tag}_A"]).max() < 1e-5
        assert np.abs(s - g[f"{tag}_start"]).max() < 1e-4 and np.abs(e - g[f"{tag}_end"]).max() < 1e-4


def test_feature_gather_reference_golden(oracle):
    g = load_golden("feature_gather_ref.npz")
    assert np.abs(oracle.feature_gather(g["fmap"], g["xy"]) - g["out"]).max() < 1e-5


# ------------------------------------------------------------------ C ABI surface (no GPU needed)
def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "jmodt_hip.h")).read()
    return sorted(set(re.findall(r"\b(jm_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from jmodt_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 24
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/jmodt_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == declared, "jmodt_amd/_lib.py SIGNATURES out of sync with the header"
    assert _lib.load().jm_version() >= 100


def test_ops_refuse_cpu_tensors():
    """the product path has no CPU fallback: GPU ops must fail loudly on CPU tensors"""
    import torch
    from jmodt_amd.ops.pointnet2 import pointnet2_utils as pu
    from jmodt_amd.ops.iou3d import iou3d_utils
    from jmodt_amd.ops.roipool3d import roipool3d_utils
    from jmodt_amd.ops import fusion
    x = torch.zeros(1, 16, 3)
    with pytest.raises(RuntimeError, match="GPU tensor"):
        pu.farthest_point_sample(x, 4)
    with pytest.raises(RuntimeError, match="GPU tensor"):
        pu.ball_query(0.5, 4, x, x[:, :4].contiguous())
    with pytest.raises(RuntimeError, match="GPU tensor"):
        iou3d_utils.boxes_iou_bev(torch.zeros(2, 5), torch.zeros(2, 5))
    with pytest.raises(RuntimeError, match="GPU tensor"):
        roipool3d_utils.roipool3d_gpu(x, torch.zeros(1, 16, 2), torch.zeros(1, 2, 7), 0.2, 8)
    with pytest.raises(RuntimeError, match="GPU"):
        fusion.feature_gather(torch.zeros(1, 4, 8, 8), torch.zeros(1, 5, 2))


def test_invalid_arguments_return_error_codes():
    """the C ABI reports errors instead of exit()ing (no GPU touched: validation comes first)"""
    from jmodt_amd import _lib
    lib = _lib.load()
    assert lib.jm_ball_query(1, 16, 4, 0.5, 0, None, None, None, None) == 1  # nsample = 0
    assert b"nsample" in lib.jm_last_error()
    assert lib.jm_furthest_point_sampling(-1, 16, 4, None, None, None, None) == 1
    assert lib.jm_nms_workspace_bytes(6300) == 6300 * 99 * 8
    assert lib.jm_nms_workspace_bytes(0) == 0


def test_cpu_entry_points_match_reference_golden():
    """pts_in_boxes3d_cpu / roipool3d_cpu are CPU functions in the reference API as well
    (roipool3d.cpp:97-195); the product's host implementation must reproduce the reference's."""
    import torch
    from jmodt_amd.ops.roipool3d import roipool3d_utils as ru
    g = load_golden("roipool3d_ref.npz")
    S = int(g["S"])
    for b in range(2):
        pts, eb, feat = (torch.from_numpy(g[k][b]) for k in ("pts", "enlarged", "feat"))
        masks = ru.pts_in_boxes3d_cpu(pts, eb)
        assert np.array_equal(torch.stack(masks).numpy().astype(np.uint8), g["flags"][b])
        pp, pf, ef = ru.roipool_pc_cpu(pts, feat, eb, S)
        assert np.array_equal(pp.numpy(), g["pooled"][b, :, :, :3])
        assert np.array_equal(pf.numpy(), g["pooled"][b, :, :, 3:])
        assert np.array_equal(ef.numpy().astype(np.int32), g["empty"][b])


def test_state_dict_names_match_reference_layout():
    from jmodt_amd.ops import affinity
    from jmodt_amd.ops.pointnet2.pointnet2_modules import PointnetFPModule, PointnetSAModuleMSG
    head = affinity.make_affinity_mlp()
    assert list(head.state_dict()) == ["0.conv.weight", "0.conv.bias", "2.conv.weight", "2.conv.bias",
                                      "3.conv.weight", "3.conv.bias"]
    assert head[0].conv.weight.shape == (512, 512, 1) and head[3].conv.weight.shape == (1, 512, 1)
    assert sum(p.numel() for p in head.parameters()) == 525825
    sa = PointnetSAModuleMSG(npoint=64, radii=[1.0, 2.0], nsamples=[8, 16], mlps=[[6, 16, 32], [6, 16, 32]])
    keys = list(sa.state_dict())
    assert "mlps.0.layer0.conv.weight" in keys and "mlps.1.layer1.bn.bn.running_mean" in keys
    assert sa.mlps[0].layer0.conv.weight.shape == (16, 9, 1, 1)   # use_xyz adds 3 input channels
    fp = PointnetFPModule(mlp=[32, 16])
    assert "mlp.layer0.conv.weight" in fp.state_dict()


def test_layer_builders_match_reference_modules():
    """pytorch_utils mirror vs the REFERENCE's own SharedMLP / Conv1d / FC (pytorch_utils.py:6-205), eval mode,
    random BatchNorm statistics: the reference state_dict loads by name and the outputs agree"""
    import torch
    from jmodt_amd.ops.pointnet2 import pytorch_utils as pt
    g = load_golden("layer_builders_ref.npz")
    mods = {"mlp": pt.SharedMLP([9, 16, 24, 40], bn=True), "conv1d": pt.Conv1d(12, 20, bn=True), "fc": pt.FC(10, 6, bn=True)}
    for tag, mod in mods.items():
        sd = {k[len(tag) + 1:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(tag + ".")}
        missing, unexpected = mod.load_state_dict(sd, strict=True)
        assert not missing and not unexpected
        mod.eval()
        with torch.no_grad():
            y = mod(torch.from_numpy(g[f"x_{tag}"])).numpy()
        assert np.abs(y - g[f"y_{tag}"]).max() < 1e-5, tag


def test_fold_shared_mlp_equals_eval_forward():
    """BN folding used by the fused SA kernel == the module's eval-mode forward (on the reference's weights)"""
    import torch
    from jmodt_amd.ops.pointnet2 import pytorch_utils as pt
    from jmodt_amd.ops.pointnet2.fused import fold_shared_mlp
    g = load_golden("layer_builders_ref.npz")
    mlp = pt.SharedMLP([9, 16, 24, 40], bn=True)
    mlp.load_state_dict({k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("mlp.")})
    mlp.eval()
    x = torch.from_numpy(g["x_mlp"])                       # (B, C, H, W)
    h = x
    for W, b in fold_shared_mlp(mlp):
        h = torch.relu(torch.einsum("oc,bchw->bohw", W, h) + b[None, :, None, None])
    assert (h - torch.from_numpy(g["y_mlp"])).abs().max().item() < 1e-5


def test_canonical_transform_and_boxes_dist_vs_reference_geometry(oracle):
    """the oracle's canonical transformation (used by roipool3d_canonical) vs the reference's own
    rotate_pc_along_y_torch, and oracle.boxes_dist vs data_association.py:10-28 evaluated with the reference's
    corner function (tests/golden/make_golden.py)"""
    g = load_golden("geometry_ref.npz")
    rois, xyz = g["rois"], g["pooled_xyz"]
    f = np.float32
    c = xyz - rois[:, None, 0:3]
    cosa, sina = np.cos(rois[:, 6]).astype(f)[:, None], np.sin(rois[:, 6]).astype(f)[:, None]
    x, z = c[..., 0].copy(), c[..., 2].copy()
    c[..., 0] = x * cosa + z * (-sina)
    c[..., 2] = x * sina + z * cosa
    assert np.abs(c - g["canonical"]).max() < 1e-5
    # the same through the oracle entry: one point per "cloud slot" that lies inside its RoI is pooled unchanged
    pts = rois[None, :, 0:3].copy()
    pts[..., 1] -= rois[None, :, 3] / 2                                 # box centres (y = bottom - h/2): inside
    got, flag = oracle.roipool3d_canonical(pts, np.zeros((1, len(rois), 1), f), rois[None], 0.0, 4)
    assert not flag.any()
    want0 = np.zeros(3, f)
    for i, r in enumerate(rois):
        assert np.abs(got[0, i, 0, 0] - want0[0]).max() < 1e-5 and np.abs(got[0, i, 0, 2]).max() < 1e-5
        assert abs(got[0, i, 0, 1] + r[3] / 2) < 1e-5                   # centre is h/2 above the bottom: y' = -h/2
    assert np.abs(oracle.boxes_dist(g["boxes_a"], g["boxes_b"]) - g["boxes_dist"]).max() < 2e-5
