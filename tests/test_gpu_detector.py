"""GPU tier (-m gpu): the COMPOSED detect + affinity forward (jmodt_amd/detector.py, BASELINE configs[2] + affinity)
against the chained CPU oracle (oracle/pipeline.py), plus the detection post-processing / hand-off entry points
 and smoke-sized runs of every bench.py workload (configs[3] included).

Chained parity is checked the way a sequential pipeline with discrete decisions has to be checked:
  * free-running for the continuous part (backbone + RPN heads): GPU vs float64 oracle from the same inputs;
  * teacher-forced across discrete decisions: each later oracle stage consumes the GPU's previous stage, so a
    1e-6 score difference cannot turn into a different RoI set and hide (or fake) an error downstream.
Bars: index / selection outputs bit-exact, float outputs 1e-4 (BASELINE.json), scaled by the tensor's magnitude
where activations exceed 1.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from jmodt_amd import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def close(got, want, tol=1e-4):
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    want = want.detach().cpu().numpy() if isinstance(want, torch.Tensor) else np.asarray(want)
    assert got.shape == want.shape, (got.shape, want.shape)
    scale = max(1.0, float(np.abs(want).max()))
    err = float(np.abs(got.astype(np.float64) - want.astype(np.float64)).max())
    assert err <= tol * scale, f"max |diff| {err:.3e} > {tol} * {scale:.3g}"


def tiny_frames(B, N, seed):
    """dense little scene so that balls / RoIs actually contain neighbours"""
    rng = np.random.default_rng(seed)
    lo, hi = np.array([-8.0, -1.0, 2.0], np.float32), np.array([8.0, 1.0, 18.0], np.float32)
    xyz = (rng.random((B, N, 3), dtype=np.float32) * (hi - lo) + lo).astype(np.float32)
    xyz[:, N - N // 10:] = xyz[:, :N // 10]          # exact duplicates, as kitti_dataset.py:243-247 produces
    img = synth.image(B, seed + 1, 96, 320, native=(94, 310))
    xy = synth.pts_xy(xyz, 320, 96)
    return xyz, img, xy


def make_engine(seed=0, cfg=None, conv_find=False):
    """conv_find off by default HERE: several tests compare two forwards bit for bit, and the kernels MIOpen's find mode picks
    for the image convolutions are split-K (atomic adds: 1e-7 run to run); the full-width fixtures, whose comparisons carry a
    tolerance, run with the engine's default (on)"""
    from jmodt_amd.detector import DetectAffinityEngine, DetectorConfig
    torch.manual_seed(seed)
    eng = DetectAffinityEngine(cfg or DetectorConfig.tiny())
    eng.conv_find = conv_find
    g = torch.Generator().manual_seed(seed + 1)
    for m in eng.modules():     # non-trivial BatchNorm statistics, non-zero biases, larger head weights
        if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
            m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
            m.weight.data.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
            m.bias.data.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
        elif isinstance(m, (torch.nn.Conv1d, torch.nn.Conv2d, torch.nn.Linear)) and m.bias is not None:
            m.bias.data.copy_(torch.randn(m.bias.shape, generator=g) * 0.05)
    with torch.no_grad():       # make the heads say something: scores around 0, visible box regression
        eng.rpn.rpn_cls_layer[2].conv.bias.zero_()
        eng.rpn.rpn_cls_layer[2].conv.weight.mul_(8.0)
        eng.rpn.rpn_reg_layer[2].conv.weight.copy_(torch.randn(eng.rpn.rpn_reg_layer[2].conv.weight.shape, generator=g) * 0.3)
        eng.rcnn_net.reg_layer[-1].conv.weight.copy_(torch.randn(eng.rcnn_net.reg_layer[-1].conv.weight.shape, generator=g) * 0.3)
        eng.rcnn_net.cls_layer[-1].conv.weight.mul_(6.0)
    return eng.eval()


@pytest.fixture(scope="module")
def run():
    from oracle.pipeline import Chain
    eng = make_engine().to(DEV)
    xyz, img, xy = tiny_frames(2, 2048, 17)
    with torch.no_grad():
        cache, aff, inter = eng(T(xyz), T(img), T(xy))
    torch.cuda.synchronize()
    chain = Chain(eng.state_dict(), eng.cfg, torch.float64)
    return dict(eng=eng, xyz=xyz, img=img, xy=xy, cache=cache, aff=aff, inter=inter, chain=chain)


def test_backbone_and_rpn_heads_free_running(run):
    want = run["chain"].rpn(run["xyz"], run["img"], run["xy"])
    inter = run["inter"]
    for lv, (got_idx, want_idx) in enumerate(zip(run["eng"].last_fps_idx, run["chain"].last["fps_idx"])):
        assert np.array_equal(got_idx.cpu().numpy(), want_idx), f"FPS level {lv + 1}"
    close(inter["backbone_features"], want["backbone_features"])
    close(inter["rpn_cls"], want["rpn_cls"])
    close(inter["rpn_reg"], want["rpn_reg"])


def test_proposals_roipool_rcnn_teacher_forced(run, oracle):
    inter, chain, cfg = run["inter"], run["chain"], run["eng"].cfg
    rpn_cls, rpn_reg = inter["rpn_cls"].cpu().numpy(), inter["rpn_reg"].cpu().numpy()
    feats = inter["backbone_features"].cpu().numpy()
    # proposal layer: the oracle selects from the GPU's decoded boxes (decode itself is compared at 1e-4)
    from jmodt_amd.ops.proposal import decode_rpn_proposals
    dec = decode_rpn_proposals(T(run["xyz"]), inter["rpn_reg"], cfg.rpn_loc_scope, cfg.rpn_loc_bin_size,
                               cfg.rpn_num_head_bin, cfg.mean_size).cpu().numpy()
    close(dec, oracle.decode_rpn_proposals(run["xyz"], rpn_reg, cfg.rpn_loc_scope, cfg.rpn_loc_bin_size,
                                           cfg.rpn_num_head_bin, cfg.mean_size))
    wb, ws = oracle.proposal_select(rpn_cls[:, :, 0], dec, cfg.rpn_pre_nms_top_n, cfg.rpn_post_nms_top_n,
                                    cfg.rpn_nms_thresh, cfg.rpn_nms_type)
    rois = inter["rois"].cpu().numpy()
    assert np.array_equal(rois, wb) and np.array_equal(inter["roi_scores_raw"].cpu().numpy(), ws)
    assert (ws != 0).sum() >= cfg.rpn_post_nms_top_n      # the scene fills every RoI slot
    # RoI pooling + canonical transform on the GPU's RoIs / features
    want_pts, _ = chain.roi_pool(run["xyz"], rpn_cls, feats, rois)
    got_pts = inter["pts_input"].cpu().numpy()
    assert np.array_equal(got_pts[..., 3], want_pts[..., 3]) and np.array_equal(got_pts[..., 5:], want_pts[..., 5:])
    close(got_pts[..., 4], want_pts[..., 4], 1e-6)      # the depth channel: torch.norm vs numpy, 1 ulp apart
    close(got_pts[..., :3], want_pts[..., :3])
    assert (np.abs(got_pts[..., 5:]).sum(axis=(1, 2)) > 0).mean() > 0.5      # most RoIs are not empty
    # RCNN on the GPU's pooled points
    want = chain.rcnn(got_pts)
    close(inter["rcnn_feat"], want["rcnn_feat"])
    close(inter["rcnn_cls"], want["rcnn_cls"])
    close(inter["rcnn_reg"], want["rcnn_reg"])


def test_detections_and_affinity_teacher_forced(run, oracle):
    inter, chain, cfg, cache = run["inter"], run["chain"], run["eng"].cfg, run["cache"]
    rois = inter["rois"].cpu().numpy()
    B, M = rois.shape[:2]
    reg = inter["rcnn_reg"].cpu().numpy()
    boxes = inter["pred_boxes3d"].cpu().numpy()
    close(boxes, oracle.decode_rcnn_boxes(rois.reshape(-1, 7), reg, cfg.rcnn_loc_scope, cfg.rcnn_loc_bin_size,
                                          cfg.rcnn_num_head_bin, cfg.mean_size).reshape(B, M, 7))
    raw = inter["rcnn_cls"].cpu().numpy().reshape(B, M)
    keep = oracle.select_detections(boxes, raw, cfg.rcnn_score_thresh, cfg.rcnn_nms_thresh)
    counts = cache.counts_host()
    assert sum(counts) > 0
    feats = inter["rcnn_feat"].view(B, M, -1).cpu().numpy()
    for b in range(B):
        assert counts[b] == len(keep[b])
        assert np.array_equal(cache.roi_index[b, :counts[b]].cpu().numpy(), keep[b])
        bx, sc, ft = cache.to_host(b)
        assert np.array_equal(bx, boxes[b][keep[b]]) and np.array_equal(ft, feats[b][keep[b]])
        assert (cache.boxes[b, counts[b]:] == 0).all() and (cache.feats[b, counts[b]:] == 0).all()
    f64 = torch.from_numpy(feats).double()
    for b in range(B):
        A, s, e = run["aff"][b]
        wA, ws, we = chain.affinity(f64[b - 1], f64[b])
        close(A, wA); close(s, ws); close(e, we)


# ------------------------------------------------------------------------------------------------------------------
# the BENCHMARKED network: full widths (point_rcnn.py:24-70 with config.py:71-139 shapes: 16.7 M parameters, hidden
# widths to 512), 16384-point frames on the 384x1280 canvas, 128 RoIs x 512 points — the configuration bench.py
# times selects different kernels than DetectorConfig.tiny() (sa_mlp_pm C = 128, sa_mlp_wide hidden 512, rcnn_lift
# with the hoisted layer, rocBLAS at LI-Fusion level 4, the 128-RoI affinity batch)
# ------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module", params=["configs2", "configs4", "reference100", "kitti", "packed"])
def full_run(request):
    """configs2: the headline workload's shapes (16384 points, 128 RoIs per frame); configs4: BASELINE configs[4] through the
    SAME composed engine (65536 points per frame: co-operative FPS, hash-grid ball query and 3-NN at the first level;
    256 RoIs, 256 x 256 affinity); reference100: DetectorConfig() = the reference's own TEST configuration, 100 RoIs per frame
    (config.py:204,213: RoI counts and affinity sizes that are no multiple of any tile); kitti: the headline shapes on the KITTI-like
    cloud (density ~ 1/z, ground plane + object clusters: RoIs with hundreds of distinct points, other compaction patterns); packed: every
    point inside one of 16 car-sized boxes (RoIs of >= 512 distinct points: nothing to compact, dense neighbourhoods everywhere)"""
    import dataclasses
    from jmodt_amd.detector import DetectorConfig
    from jmodt_amd.profile import prof
    from oracle.pipeline import Chain
    dense = request.param == "configs4"
    cfg = dataclasses.replace(DetectorConfig.survey(), rpn_post_nms_top_n=256) if dense else DetectorConfig.survey()
    if request.param == "reference100":
        cfg = DetectorConfig()
    eng = make_engine(seed=5, cfg=cfg, conv_find=True).to(DEV)
    xyz, img, xy = synth.frames(2, 65536 if dense else 16384, 4321, kind=request.param if request.param in ("kitti", "packed") else "uniform")
    with torch.no_grad():
        eng(T(xyz), T(img), T(xy))                         # warm-up: packs / folds every weight
        prof.reset()
        prof.enabled = True
        try:
            cache, aff, inter = eng(T(xyz), T(img), T(xy))
            torch.cuda.synchronize()
            taken = set(prof.records)
        finally:
            prof.enabled = False
            prof.reset()
    chain = Chain(eng.state_dict(), eng.cfg, torch.float64)
    return dict(eng=eng, xyz=xyz, img=img, xy=xy, cache=cache, aff=aff, inter=inter, chain=chain, taken=taken)


def test_full_width_kernel_selection(full_run):
    """the entries the benchmarked configuration takes (from the profiler's records of the compared run itself)"""
    taken, eng = full_run["taken"], full_run["eng"]
    names = " ".join(sorted(taken))
    for needle in ("rcnn_sa1/sa_mlp_pm_forward", "rcnn_sa2/sa_mlp_pm_forward", "rcnn_sa3/sa_mlp_forward", "rpn_sa2/sa_mlp_pm_forward",
                   "rpn_sa3/sa_mlp_forward", "rpn_sa4/sa_mlp_forward", "rcnn_lift_forward", "conv1d_stack_forward",
                   "li_fusion_final/image_fusion_gather", "li_fusion1/attention_fusion_forward", "li_fusion3/attention_fusion_forward",
                   "li_fusion_final/attention_fusion_forward", f"affinity_2x{eng.cfg.rpn_post_nms_top_n}x{eng.cfg.rpn_post_nms_top_n}/affinity_forward_batched", "linear_rows",
                   "conv3x3_rgb_bias_relu", "conv3x3_wino_bias_relu", "proposal_layer/", "roipool3d_canonical", "detections/nms_batched",
                   "fps_pyramid/L1/furthest_point_sampling_xyz", "three_nn", "three_interpolate"):
        assert needle in names, (needle, names)
    # the RPN set-abstraction scales take the duplicate-aware (listed) form of their kernels, planned once per level
    for needle in ("rpn_sa1/sa_mlp_forward_listed", "rpn_sa2/sa_mlp_pm_forward_listed", "rpn_sa3/sa_mlp_forward_listed",
                   "rpn_sa4/sa_mlp_forward_listed", "rpn_sa1/sa_group_plan_dual", "rpn_sa4/sa_group_plan_dual"):
        assert needle in names, (needle, names)
    assert "li_fusion4/attention_fusion_forward" not in names          # 512 rows x 1024 channels: rocBLAS GEMMs
    assert eng._folded["rcnn_lift"].ho == 128                           # lift kernel carries RCNN SA1's hoisted first layer
    # the `wide` variant is what jm_sa_mlp_forward dispatches to at hidden widths > 128 / GroupAll
    import ctypes
    from jmodt_amd import _lib
    lib = _lib.load()
    for (b, n, m, c, ns, ga, widths) in ((2, 1024, 256, 256, 16, 0, [259, 128, 196, 256]), (2, 256, 64, 512, 32, 0, [515, 256, 384, 512]),
                                         (256, 32, 1, 256, 32, 1, [259, 256, 256, 512])):
        assert lib.jm_sa_mlp_supported(b, n, m, c, ns, ga, 3, (ctypes.c_int * 4)(*widths)) == 2, widths


def test_full_width_backbone_and_rpn_heads_free_running(full_run):
    want = full_run["chain"].rpn(full_run["xyz"], full_run["img"], full_run["xy"])
    inter = full_run["inter"]
    for lv, (got_idx, want_idx) in enumerate(zip(full_run["eng"].last_fps_idx, full_run["chain"].last["fps_idx"])):
        assert np.array_equal(got_idx.cpu().numpy(), want_idx), f"FPS level {lv + 1}"
    assert float(want["backbone_features"].abs().max()) > 0.5
    close(inter["backbone_features"], want["backbone_features"])
    close(inter["rpn_cls"], want["rpn_cls"])
    close(inter["rpn_reg"], want["rpn_reg"])


def test_full_width_proposals_roipool_rcnn_teacher_forced(full_run, oracle):
    inter, chain, cfg = full_run["inter"], full_run["chain"], full_run["eng"].cfg
    xyz = full_run["xyz"]
    rpn_cls, rpn_reg = inter["rpn_cls"].cpu().numpy(), inter["rpn_reg"].cpu().numpy()
    feats = inter["backbone_features"].cpu().numpy()
    from jmodt_amd.ops.proposal import decode_rpn_proposals
    dec = decode_rpn_proposals(T(xyz), inter["rpn_reg"], cfg.rpn_loc_scope, cfg.rpn_loc_bin_size,
                               cfg.rpn_num_head_bin, cfg.mean_size).cpu().numpy()
    close(dec, oracle.decode_rpn_proposals(xyz, rpn_reg, cfg.rpn_loc_scope, cfg.rpn_loc_bin_size, cfg.rpn_num_head_bin, cfg.mean_size))
    wb, ws = oracle.proposal_select(rpn_cls[:, :, 0], dec, cfg.rpn_pre_nms_top_n, cfg.rpn_post_nms_top_n,
                                    cfg.rpn_nms_thresh, cfg.rpn_nms_type)
    rois = inter["rois"].cpu().numpy()
    M = cfg.rpn_post_nms_top_n
    assert rois.shape == (2, M, 7)
    assert np.array_equal(rois, wb) and np.array_equal(inter["roi_scores_raw"].cpu().numpy(), ws)
    assert (np.abs(rois).sum(-1) > 0).sum() >= 2 * M * 0.78  # the proposal layer fills (nearly) every RoI slot
    want_pts, _ = chain.roi_pool(xyz, rpn_cls, feats, rois)
    got_pts = inter["pts_input"].cpu().numpy()
    assert got_pts.shape == (2 * M, 512, 133)
    assert np.array_equal(got_pts[..., 3], want_pts[..., 3]) and np.array_equal(got_pts[..., 5:], want_pts[..., 5:])
    close(got_pts[..., 4], want_pts[..., 4], 1e-6)
    close(got_pts[..., :3], want_pts[..., :3])
    assert (np.abs(got_pts[..., 5:]).sum(axis=(1, 2)) > 0).mean() > 0.5
    want = chain.rcnn(got_pts)                              # 256 RoIs x 512 points, un-fused, float64
    close(inter["rcnn_feat"], want["rcnn_feat"])
    close(inter["rcnn_cls"], want["rcnn_cls"])
    close(inter["rcnn_reg"], want["rcnn_reg"])


def test_full_width_detections_and_affinity_teacher_forced(full_run, oracle):
    inter, chain, cfg, cache = full_run["inter"], full_run["chain"], full_run["eng"].cfg, full_run["cache"]
    rois = inter["rois"].cpu().numpy()
    B, M = rois.shape[:2]
    reg = inter["rcnn_reg"].cpu().numpy()
    boxes = inter["pred_boxes3d"].cpu().numpy()
    close(boxes, oracle.decode_rcnn_boxes(rois.reshape(-1, 7), reg, cfg.rcnn_loc_scope, cfg.rcnn_loc_bin_size,
                                          cfg.rcnn_num_head_bin, cfg.mean_size).reshape(B, M, 7))
    raw = inter["rcnn_cls"].cpu().numpy().reshape(B, M)
    keep = oracle.select_detections(boxes, raw, cfg.rcnn_score_thresh, cfg.rcnn_nms_thresh)
    counts = cache.counts_host()
    assert sum(counts) > 0
    feats = inter["rcnn_feat"].view(B, M, -1).cpu().numpy()
    for b in range(B):
        assert counts[b] == len(keep[b])
        assert np.array_equal(cache.roi_index[b, :counts[b]].cpu().numpy(), keep[b])
        bx, sc, ft = cache.to_host(b)
        assert np.array_equal(bx, boxes[b][keep[b]]) and np.array_equal(ft, feats[b][keep[b]])
    f64 = torch.from_numpy(feats).double()
    for b in range(B):                                      # 128 x 128 pairs x 512 channels per frame
        A, s, e = full_run["aff"][b]
        wA, ws_, we = chain.affinity(f64[b - 1], f64[b])
        close(A, wA); close(s, ws_); close(e, we)


def test_engine_without_side_streams_gives_identical_results(run):
    eng = run["eng"]
    eng.overlap = False
    try:
        with torch.no_grad():
            cache, aff, inter = eng(T(run["xyz"]), T(run["img"]), T(run["xy"]))
    finally:
        eng.overlap = True
    for k in ("backbone_features", "rpn_reg", "rois", "pts_input", "rcnn_feat", "pred_boxes3d"):
        assert torch.equal(inter[k], run["inter"][k]), k
    assert torch.equal(cache.count, run["cache"].count)
    assert torch.equal(aff[1][0], run["aff"][1][0]) and torch.equal(aff[0][1], run["aff"][0][1])   # (no float atomics anywhere)


def test_next_batch_prefetch_gives_identical_results(run):
    """announcing the next batch starts its FPS pyramid early; results must not change, a stale announcement for a
    different tensor must be dropped"""
    eng = run["eng"]
    a, img, xy = T(run["xyz"]), T(run["img"]), T(run["xy"])
    other = T(run["xyz"][:, ::-1].copy())
    with torch.no_grad():
        eng(a, img, xy, next_xyz=a)                       # announces `a`
        _, _, i1 = eng(a, img, xy, next_xyz=other)        # consumes the announced pyramid, announces `other`
        _, _, i2 = eng(a, img, xy)                        # `other` was announced, `a` arrives: dropped, recomputed
    for k in ("backbone_features", "rois", "rcnn_feat"):
        assert torch.equal(i1[k], run["inter"][k]) and torch.equal(i2[k], run["inter"][k]), k
    assert eng._prefetched == []


def test_two_fps_pyramids_in_flight_give_identical_results(run):
    """prefetch_depth 2: the two upcoming clouds are announced as a list, each FPS chain on a side stream of its own; one
    pyramid is started per step in steady state, results do not change, and an announcement that does not come true is dropped"""
    eng = run["eng"]
    a, img, xy = T(run["xyz"]), T(run["img"]), T(run["xy"])
    b = a.clone()
    other = T(run["xyz"][:, ::-1].copy())
    eng.prefetch_depth = 2
    try:
        with torch.no_grad():
            n0 = eng._fps_launches
            eng(a, img, xy, next_xyz=[b, a], next_image=img)            # starts b and a
            assert eng._fps_launches == n0 + 2 and [x is y for (x, _), y in zip(eng._prefetched, (b, a))] == [True, True]
            assert eng._prefetched[0][1]._side != eng._prefetched[1][1]._side
            _, _, i1 = eng(b, img, xy, next_xyz=[a, b], next_image=img)  # consumes b; a is in flight already: starts b only
            assert eng._fps_launches == n0 + 3
            _, _, i2 = eng(a, img, xy, next_xyz=[b, other])              # consumes a; b in flight, starts other
            assert eng._fps_launches == n0 + 4
            _, _, i3 = eng(b, img, xy, next_xyz=[a])                     # consumes b; `other` does not come true: dropped, a started
            assert eng._fps_launches == n0 + 5 and len(eng._prefetched) == 1
            _, _, i4 = eng(a, img, xy)
            torch.cuda.synchronize()
    finally:
        eng.prefetch_depth = 1
    for k in ("backbone_features", "rois", "rcnn_feat"):
        for got in (i1, i2, i3, i4):
            assert torch.equal(got[k], run["inter"][k]), k
    assert eng._prefetched == []


@pytest.mark.parametrize("late", [False, True])
def test_next_image_prefetch_gives_identical_results(run, late):
    """announcing the next batch's image starts its image pyramid on the side stream (under this batch's backbone, or
    after it); results must not change, an announcement for a different image must be dropped"""
    eng = run["eng"]
    a, img, xy = T(run["xyz"]), T(run["img"]), T(run["xy"])
    other = T(run["img"][:, :, ::-1].copy())
    eng.prefetch_image_late = late
    try:
        with torch.no_grad():
            eng(a, img, xy, next_xyz=a, next_image=img)
            assert eng._prefetched_img is not None and eng._prefetched_img[0] is img
            _, _, i1 = eng(a, img, xy, next_xyz=a, next_image=other)      # consumes the announced pyramid
            _, _, i2 = eng(a, img, xy)                                   # `other` was announced, `img` arrives: dropped
            torch.cuda.synchronize()
    finally:
        eng.prefetch_image_late = False
    for k in ("backbone_features", "rois", "rcnn_feat"):
        assert torch.equal(i1[k], run["inter"][k]) and torch.equal(i2[k], run["inter"][k]), k
    assert eng._prefetched_img is None


def test_unfused_sa_path_matches_fused(run):
    """the same engine with the fused SA kernel disabled (QueryAndGroup + GroupAll + torch convs) — a8 / a5"""
    eng = run["eng"]
    mods = [m for m in eng.modules() if hasattr(m, "fuse")]
    for m in mods:
        m.fuse = False
    try:
        with torch.no_grad():
            rpn_out = eng.rpn_forward(T(run["xyz"]), T(run["img"]), T(run["xy"]))
            out = eng.rcnn_forward(run["inter"]["pts_input"])
    finally:
        for m in mods:
            m.fuse = True
    close(rpn_out["backbone_features"], run["inter"]["backbone_features"])
    close(out["rcnn_feat"], run["inter"]["rcnn_feat"])


def test_detection_cache_association_matches_oracle(run, oracle):
    """§8f-4: affinity + association cost straight from the resident cache vs the oracle on the host copies"""
    cache, eng = run["cache"], run["eng"]
    counts = cache.counts_host()
    link, se = eng.rcnn_net.link_layer, eng.rcnn_net.se_layer

    def w(h):
        return tuple(a.detach().cpu().numpy().copy() for a in (
            h[0].conv.weight[..., 0], h[0].conv.bias, h[2].conv.weight[..., 0], h[2].conv.bias,
            h[3].conv.weight.reshape(-1), h[3].conv.bias))
    res = cache.associate(0, 1, link, se, 0.6, 0.3, 0.1)
    if counts[0] == 0 or counts[1] == 0:
        assert res is None
        return
    cost, A, s, e = res
    pb, _, pf = cache.to_host(0)
    db, _, df = cache.to_host(1)
    wA, ws, we = oracle.affinity(pf, df, w(link), w(se))
    close(A, wA); close(s, ws); close(e, we)
    close(cost, oracle.association_cost(pb, db, wA, 0.6, 0.3, 0.1))


def test_decode_rcnn_boxes_vs_oracle(oracle):
    from jmodt_amd.ops.detections import decode_rcnn_boxes
    rng = np.random.default_rng(5)
    P = 1000
    rois = synth.proposals(synth.cloud(1, 4096, 3), P, 4)[0]
    reg = rng.normal(0, 1.2, (P, 46)).astype(np.float32)
    for avg in (True, False):
        got = decode_rcnn_boxes(T(rois), T(reg), avg_by_bin=avg)
        close(got, oracle.decode_rcnn_boxes(rois, reg, avg_by_bin=avg))
    assert decode_rcnn_boxes(T(rois[:0]), T(reg[:0])).shape == (0, 7)


@pytest.mark.parametrize("B,Na,Nb", [(4, 512, 20), (2, 64, 64), (3, 17, 1), (1, 1, 33)])
def test_boxes_iou3d_batched_vs_oracle(oracle, B, Na, Nb):
    """§8f-3b: the RoI sampler's per-frame boxes_iou3d_gpu loop as one launch, zero-padded ground truth"""
    from jmodt_amd.ops.detections import boxes_iou3d_batched
    from jmodt_amd.ops.iou3d.iou3d_utils import boxes_iou3d_gpu
    pts = synth.cloud(B, 2048, 9)
    gt = synth.proposals(pts, Nb, 10)
    rng = np.random.default_rng(11)
    rois = np.repeat(gt, (Na + Nb - 1) // Nb, axis=1)[:, :Na].copy()
    rois[..., :3] += rng.normal(0, 0.4, rois[..., :3].shape).astype(np.float32)
    rois[..., 6] += rng.normal(0, 0.3, rois[..., 6].shape).astype(np.float32)
    counts = rng.integers(1, Nb + 1, B).astype(np.int32)
    for b in range(B):
        gt[b, counts[b]:] = 0
    got = boxes_iou3d_batched(T(rois), T(gt), T(counts)).cpu().numpy()
    assert got.shape == (B, Na, Nb)
    for b in range(B):
        k = counts[b]
        want = oracle.boxes_iou3d(rois[b], gt[b, :k])
        assert np.abs(got[b, :, :k] - want).max() < 1e-5
        assert (got[b, :, k:] == 0).all()
        single = boxes_iou3d_gpu(T(rois[b]), T(gt[b, :k])).cpu().numpy()       # the per-frame API it replaces
        assert np.abs(got[b, :, :k] - single).max() < 1e-6
    assert got.max() > (0.3 if Na * Nb > 64 else 0.0)
    full = boxes_iou3d_batched(T(rois), T(gt)).cpu().numpy()                  # counts = None: every column valid
    assert np.array_equal(full[:, :, :1], got[:, :, :1])


def test_select_detections_synthetic(oracle):
    """score threshold + rotated NMS for a batch with clustered boxes, empty frames and full frames"""
    from jmodt_amd.ops.detections import select_detections
    rng = np.random.default_rng(2)
    B, M, C = 5, 128, 32
    pts = synth.cloud(B, 512, 1)
    base = synth.proposals(pts, 16, 2)
    boxes = np.repeat(base, M // 16, axis=1)
    boxes[..., :3] += rng.normal(0, 0.5, boxes[..., :3].shape).astype(np.float32)
    raw = rng.permutation(B * M).reshape(B, M).astype(np.float32) / (B * M) * 8 - 4
    raw[1] = -9.0          # nothing passes the score threshold
    raw[2] = np.abs(raw[2]) + 1.0   # everything passes
    feats = rng.normal(size=(B, M, C)).astype(np.float32)
    cache = select_detections(T(boxes), T(raw), T(feats), 0.2, 0.1)
    keep = oracle.select_detections(boxes, raw, 0.2, 0.1)
    counts = cache.counts_host()
    assert counts[1] == 0
    for b in range(B):
        assert counts[b] == len(keep[b]) and np.array_equal(cache.roi_index[b, :counts[b]].cpu().numpy(), keep[b])
        bx, sc, ft = cache.to_host(b)
        assert np.array_equal(bx, boxes[b][keep[b]]) and np.array_equal(ft, feats[b][keep[b]])
        assert np.allclose(sc, 1 / (1 + np.exp(-raw[b][keep[b]])), atol=1e-6)


def _strict(text):
    """json.loads that refuses NaN / Infinity (what a strict parser on the driver's side would refuse)"""
    def bad(tok):
        raise ValueError(f"non-finite constant {tok} in the bench line")
    return json.loads(text, parse_constant=bad)


def _one_compact_line(stdout):
    """the stdout contract: exactly ONE line, strict JSON, at most 4 KB (round 3's 24 KB line could not be parsed from the tail of
    stdout the driver keeps), carrying the contract's keys + roofline (+ by time)"""
    lines = [ln for ln in stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, stdout[-2000:]
    assert len(lines[0].encode()) <= 4096, len(lines[0])
    r = _strict(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "roofline_by_time", "full_record"):
        assert key in r, key
    assert "workload" in r["config"] and "kernels" not in r
    if r["roofline"] is not None:
        for key in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic"):
            assert key in r["roofline"], key
    return lines[0], r


def test_bench_default_line_is_compact_and_complete():
    """the DRIVER's command line (`python bench.py --gpus 1 --steps K --warmup W`, nothing else): one strict-JSON line of at most
    4 KB with value, roofline (incl. the rocprof cross-check), roofline_by_time, cpu_baseline and the three clouds; every rate in
    the kernel table of the full record below the machine's peaks"""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    line, r = _one_compact_line(p.stdout)
    assert r["value"] > 0 and r["unit"] == "frames/s" and r["dtype"] == "f32" and r["vs_baseline"] is None
    assert r["roofline"]["bound"] in ("mfma", "hbm") and 0 < r["roofline"]["frac"] <= 1
    assert abs(r["roofline"]["frac"] - r["roofline"]["achieved"] / r["roofline"]["peak"]) < 2e-3
    assert r["roofline_by_time"]["kernel"] and r["roofline_by_time"]["ms_per_step"] > 0
    cb = r["cpu_baseline"]
    assert cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] == "port" and cb["sample"]
    assert set(r["clouds"]) >= {"uniform", "kitti", "packed"} and all(v > 0 for v in r["clouds"].values())
    assert r["clouds"]["uniform"] == r["value"] and r["no_prefetch_value"] > 0
    full = _strict(open(os.path.join(ROOT, r["full_record"])).read())
    for k in full["kernels"]:
        assert k.get("hbm_frac", 0) <= 1.0, k                      # no fraction above the roofline (round 3: FPS 2.10)
        assert k.get("mfma_frac", 0) <= 1.0 and "executed_mfma_frac" not in k, k      # (executed flops; the dense formulation's rate is `dense_equivalent_tflops`)
        if k.get("traffic_bytes_per_launch") and k["ms_per_step"] > 0:
            per_launch_s = k["ms_per_step"] / max(k["launches_per_step"], 1) * 1e-3
            assert k["traffic_bytes_per_launch"] / per_launch_s <= 8.0e12 * 1.05, k    # counter bytes / time within the HBM peak
    assert "gpu_vs_chain" in full["cpu_baseline"] and "stage_seconds" in full["cpu_baseline"]


@pytest.mark.parametrize("workload,extra", [("detect", []), ("sa", []), ("ops", []), ("dense", ["--batch", "2"]),
                                            ("train", []), ("detect", ["--no-overlap"]), ("dense_detect", ["--batch", "2"])])
def test_bench_workloads_smoke(workload, extra):
    """every bench.py workload end to end at smoke size (configs[3]'s training step included): one JSON line with
    the contract's keys; `detect` must list the jm entry points of the whole composed path"""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline"] + (["--tiny"] if workload != "sa" else []) + extra
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    line, r = _one_compact_line(p.stdout)
    assert r["value"] > 0 and r["steps"] == 2 and r["n_gpus"] == 1
    full = json.load(open(os.path.join(ROOT, r["full_record"])))         # the kernel table lives in the full record
    assert full["value"] == r["value"] and full["roofline"]["kernel"] == r["roofline"]["kernel"]
    names = " ".join(k["kernel"] for k in full["kernels"])
    for k in full["kernels"]:                                             # a fraction is a fraction of a peak, whatever the workload
        assert k.get("mfma_frac", 0) <= 1.0 and k.get("hbm_frac", 0) <= 1.0 and "executed_mfma_frac" not in k, k
    if workload in ("detect", "train", "dense_detect"):
        for needle in ("fps_pyramid/L1/furthest_point_sampling_xyz", "rpn_sa1/", "li_fusion1/feature_gather", "three_nn",
                       "three_interpolate", "proposal_layer/", "roipool3d_canonical", "rcnn_sa1/sa_mlp_",
                       "detections/decode_rcnn_boxes", "nms_batched"):
            assert needle in names, (needle, names)
    if workload in ("detect", "dense_detect"):
        assert "affinity_forward" in names and r["roofline"] is not None
    if workload == "train":
        assert "finetune" in names


@pytest.mark.parametrize("extra", [["--workload", "train", "--tiny", "--launch"], ["--workload", "detect", "--tiny", "--launch"]])
def test_bench_self_launch_under_torch_distributed_run(extra):
    """`python bench.py --gpus N` from a plain shell re-executes itself under torch.distributed.run (the driver's N > 1
    command line); --launch takes that path on ONE GPU: rendezvous on 127.0.0.1, RCCL communicator, barriers, the MAX
    all-reduce of the timing and (train) the bucketed gradient all-reduce all run, and stdout is still exactly one JSON line"""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"] + extra
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    line, r = _one_compact_line(p.stdout)
    assert r["n_gpus"] == 1 and r["value"] > 0 and r["steps"] == 2
    assert "dp1" in r["config"]["parallelism"] or "replicas x1" in r["config"]["parallelism"]
    full_cfg = json.load(open(os.path.join(ROOT, r["full_record"])))["config"]
    assert full_cfg["rank_env"]["MIOPEN_USER_DB_PATH"].endswith(os.path.join("rank0", "db"))
    if "train" in extra:
        assert full_cfg["rank_env"]["GPU_MAX_HW_QUEUES"] == "8"      # (RCCL takes hardware queues: 381 instead of ~500 frames/s at the default 4)
        # the one-rank RCCL group takes the collective path: the gradient bucket went through all_reduce and was timed
        ga = r["grad_allreduce"]
        assert ga["issued"] == 1 and ga["ms_per_step"] > 0 and ga["bytes_per_step"] > 0 and ga["world"] == 1


def test_bench_two_ranks_through_the_drivers_launch_command():
    """the driver's N > 1 command line — `python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1
    --master-port P bench.py --gpus 2 ...` — executed on this one-GPU box with both ranks on cuda:0 (JM_BENCH_SHARE_GPU=1, a test
    hook): rendezvous, the gloo control plane (barriers, MAX of the elapsed times over ranks), rank-0-only output, per-rank MIOpen
    paths.  value = frames of BOTH ranks / the slower rank's time; the line says that it is not a benchmark"""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--tiny",
           "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["JM_BENCH_SHARE_GPU"] = "1"
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    line, r = _one_compact_line(p.stdout)                     # exactly ONE line although two ranks ran
    assert r["n_gpus"] == 2 and r["value"] > 0 and r["steps"] == 3 and r["scaling"] == "weak"
    assert "replicas x2" in r["config"]["parallelism"] and "not a benchmark" in r["config"]["parallelism"]
    assert "gloo" in r["config"]["process_groups"] and "no RCCL" in r["config"]["process_groups"]
    full = json.load(open(os.path.join(ROOT, r["full_record"])))
    assert abs(full["value"] - 2 * full["config"]["frames_per_gpu_per_step"] * 3 / (full["ms_per_step"] * 3e-3)) < 0.02 * full["value"]
    # one process per GPU: rank 0 bound to its own cores (half of the allowed ones here: both "GPUs" sit on one NUMA node or the
    # platform does not say), its own MIOpen user database / cache
    rb, envr = full["config"]["rank_binding"], full["config"]["rank_env"]
    if hasattr(os, "sched_setaffinity"):
        assert rb["pinned"] and rb["source"] in ("numa", "even-split") and 1 <= rb["n_cores"] <= max(1, len(os.sched_getaffinity(0)) // 2 + 1)
    assert envr["MIOPEN_USER_DB_PATH"].endswith(os.path.join("rank0", "db")) and envr["MIOPEN_CUSTOM_CACHE_DIR"].endswith(os.path.join("rank0", "cache"))


def test_bench_refuses_more_ranks_than_gpus():
    """--gpus 2 on a one-GPU box: the self-launched job must fail loudly (no silent fallback to one GPU)"""
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than 2 GPUs")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--tiny", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.strip().startswith("{")]
    assert "GPU(s)" in p.stderr


@pytest.mark.parametrize("ic,pc,n,B", [(64, 96, 4096, 2), (128, 256, 1024, 3), (32, 128, 16384, 1), (8, 48, 64, 2), (20, 40, 96, 1)])
def test_attention_fusion_kernel_vs_module(ic, pc, n, B):
    """csrc/li_fusion.hip against the parameter container's own forward (Linear / Conv1d / BatchNorm1d modules =
    the reference's op sequence, backbone.py:44-81) in float64 on the CPU, at the shapes of LI-Fusion levels 1, 2 and
    the final fusion (config.py:46-47) + odd widths"""
    from jmodt_amd.detector import AttentionFusion, DetectAffinityEngine, DetectorConfig
    torch.manual_seed(ic * pc)
    mod = AttentionFusion(ic, pc, pc)
    g = torch.Generator().manual_seed(1)
    for m in mod.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.2)
            m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
            m.weight.data.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
            m.bias.data.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
    mod.eval()
    P, I = torch.randn(B, pc, n, generator=g), torch.randn(B, ic, n, generator=g)
    with torch.no_grad():
        want = mod.double()(P.double(), I.double())
    mod = mod.float().to(DEV)
    eng = DetectAffinityEngine(DetectorConfig.tiny())
    with torch.no_grad():
        got = eng._attention_fusion("t", mod, P.to(DEV), I.to(DEV))
        assert "t.packed" in eng._folded                                      # the fused kernel ran
        eng.fuse_attention = False
        eng.invalidate()
        got_gemm = eng._attention_fusion("t", mod, P.to(DEV), I.to(DEV))       # rocBLAS path
    close(got, want)
    close(got_gemm, want)


def test_li_fusion_blocks_on_gpu_vs_reference_golden():
    """the reference's AttentionFusion outputs (tests/golden/fusion_ref.npz) from the fused kernel"""
    from jmodt_amd.detector import DetectAffinityEngine, DetectorConfig
    from tests.conftest import load_golden
    gd = load_golden("fusion_ref.npz")
    eng = DetectAffinityEngine(DetectorConfig.tiny())
    net = eng.rpn.backbone_net
    net.load_state_dict({k[3:]: torch.from_numpy(gd[k]) for k in gd.files if k.startswith("sd.")}, strict=False)
    eng = eng.to(DEV)
    with torch.no_grad():
        for i, mod in enumerate(net.Fusion_Conv):
            P, I = gd[f"fusion{i}_point"], gd[f"fusion{i}_img"]
            # 29 points in the golden: pad the point axis to 32 for the tile kernel, compare the first 29
            Pp, Ip = np.zeros(P.shape[:2] + (32,), np.float32), np.zeros(I.shape[:2] + (32,), np.float32)
            Pp[..., :29], Ip[..., :29] = P, I
            got = eng._attention_fusion(f"g{i}", mod, T(Pp), T(Ip))
            assert f"g{i}.packed" in eng._folded
            close(got[..., :29], gd[f"fusion{i}_out"])
        from jmodt_amd.ops.fusion import feature_gather
        close(feature_gather(T(gd["fused_map"]), T(gd["xy"])), gd["gathered"], 1e-5)
        x = T(gd["image"]).contiguous(memory_format=torch.channels_last)
        for i in range(4):                                  # BasicBlock with folded BatchNorm + the one-pass bias/ReLU kernel
            x = eng._image_block(i, x)
            close(x, gd[f"img{i + 1}"])


@pytest.mark.parametrize("full,B,N,H,W", [(False, 2, 1000, 32, 64), (False, 1, 37, 48, 160), (True, 1, 16384, 384, 1280)])
def test_sparse_image_fusion_gather_vs_dense(full, B, N, H, W):
    """csrc/image_fusion.hip (fused image feature evaluated only under the bilinear taps) vs the dense route it
    replaces: composed transposed convolutions -> (B, q, H, W) map -> feature_gather; points on pixel centres, on the
    border, outside the canvas (zeros padding) and everywhere in between"""
    from jmodt_amd.detector import DetectAffinityEngine, DetectorConfig
    from jmodt_amd.ops.fusion import PackedImageFusion, feature_gather
    torch.manual_seed(7)
    eng = DetectAffinityEngine(DetectorConfig.survey() if full else DetectorConfig.tiny())
    net = eng.rpn.backbone_net
    g = torch.Generator().manual_seed(8)
    with torch.no_grad():
        net.image_fusion_bn.running_mean.copy_(torch.randn(net.image_fusion_bn.running_mean.shape, generator=g) * 0.1)
        net.image_fusion_bn.running_var.copy_(torch.rand(net.image_fusion_bn.running_var.shape, generator=g) + 0.5)
        net.image_fusion_bn.bias.copy_(torch.randn(net.image_fusion_bn.bias.shape, generator=g) * 0.2)
        for dc in net.DeConv:
            dc.bias.copy_(torch.randn(dc.bias.shape, generator=g) * 0.1)
    eng = eng.to(DEV)
    cfg = eng.cfg
    maps = [torch.randn(B, c, H // k, W // k, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
            for c, k in zip(cfg.img_channels[1:], cfg.deconv_kernels)]
    xy = torch.rand(B, N, 2, generator=g) * 2.3 - 1.15
    xy[0, 0] = torch.tensor([-1.0, -1.0]); xy[0, 1] = torch.tensor([1.0, 1.0]); xy[0, 2] = torch.tensor([1.4, 0.1])
    xy[0, 3] = torch.tensor([2 * 5 / (W - 1) - 1, 2 * 7 / (H - 1) - 1])       # exactly on a pixel centre
    xy = xy.to(DEV)
    with torch.no_grad():
        dense = feature_gather(eng._image_fusion_map(maps), xy)
        sparse = PackedImageFusion(*eng._composed_image_fusion(), list(cfg.deconv_kernels))
        assert sparse.supported(maps, H, W)
        got = sparse(maps, xy, H, W)
        again = sparse(maps, xy, H, W)
    assert got.shape == dense.shape == (B, cfg.img_features_channel // 4, N)
    assert dense.abs().max().item() > 0.1
    close(got, dense)
    assert torch.equal(got, again)                     # deterministic whatever order the phase sort produced
    out_of_canvas = (xy.abs() > 1.0 + 2.0 / (min(H, W) - 1) + 1e-3).any(dim=2)     # more than one pixel outside
    assert out_of_canvas.any() and (got.transpose(1, 2)[out_of_canvas].abs().max().item() == 0.0)


def test_rcnn_lift_kernel_and_hoisted_first_layer(run):
    """csrc/rcnn_lift.hip (xyz_up + merge_down + hoisted first SA layer in one launch) vs the rocBLAS route and vs the
    kernel without hoisting, on the run's pooled RoI points; plus the full-size widths on random rows"""
    eng = run["eng"]
    pts = run["inter"]["pts_input"]
    with torch.no_grad():
        a = eng.rcnn_forward(pts)
        assert eng._folded["rcnn_lift"].ho > 0
        eng.fuse_rcnn_lift = False
        try:
            b = eng.rcnn_forward(pts)
        finally:
            eng.fuse_rcnn_lift = True
    for k in ("rcnn_feat", "rcnn_cls", "rcnn_reg"):
        close(a[k], b[k])
    # full-size widths (5 -> 128 -> 128, merge 256 -> 128, hoisted 131 -> 128), 512 points per RoI
    from jmodt_amd.ops.rcnn_lift import PackedRcnnLift
    g = torch.Generator().manual_seed(3)
    R, S, C = 9, 512, 128
    mk = lambda o, i: (torch.randn(o, i, generator=g) * (2.0 / i) ** 0.5).to(DEV)   # noqa: E731
    bias = lambda o: (torch.randn(o, generator=g) * 0.1).to(DEV)                    # noqa: E731
    up = [(mk(128, 5), bias(128)), (mk(128, 128), bias(128))]
    merge = (mk(128, 256), bias(128))
    W1, b1 = mk(128, 131), bias(128)
    x = torch.randn(R, S, 5 + C, generator=g).to(DEV)
    with torch.no_grad():
        rows = x.view(R * S, -1).double()
        h = torch.relu(rows[:, :5] @ up[0][0].double().t() + up[0][1].double())
        h = torch.relu(h @ up[1][0].double().t() + up[1][1].double())
        m = torch.relu(torch.cat([h, rows[:, 5:]], 1) @ merge[0].double().t() + merge[1].double())
        u = torch.cat([rows[:, :3], m], 1) @ W1.double().t() + b1.double()
        got_m = PackedRcnnLift(up, merge)(x)
        got_u = PackedRcnnLift(up, merge, (W1, b1))(x)
        got_u_pm = PackedRcnnLift(up, merge, (W1, b1))(x, point_major=True)      # (R, S, h): the layout sa_mlp_pm gathers
        assert got_u_pm.shape == (R, S, 128) and torch.equal(got_u_pm.transpose(1, 2), got_u)
    close(got_m, m.view(R, S, -1).transpose(1, 2))
    close(got_u, u.view(R, S, -1).transpose(1, 2))


@pytest.mark.parametrize("M,K,N,relu", [(1024, 512, 512, True), (1024, 512, 46, False), (1024, 512, 1, False), (37, 8, 33, True),
                                        (1, 64, 5, False), (0, 64, 5, True), (100, 520, 70, True)])
def test_linear_rows_vs_fp64(M, K, N, relu):
    """jm_linear_rows (one dense layer on plain rows, the RCNN heads' launch) vs fp64 matmul"""
    from jmodt_amd.ops.affinity import linear_rows
    g = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K, generator=g).to(DEV)
    W = (torch.randn(N, K, generator=g) * (2.0 / K) ** 0.5).to(DEV)
    b = (torch.randn(N, generator=g) * 0.1).to(DEV)
    got = linear_rows(x, W, b, relu)
    want = x.double() @ W.double().t() + b.double()
    if relu:
        want = torch.relu(want)
    assert got.shape == (M, N)
    if M:
        close(got, want)
        if not relu:
            assert (got < 0).any()
    with pytest.raises(RuntimeError):
        linear_rows(torch.randn(4, 12).to(DEV), torch.randn(3, 12).to(DEV), torch.randn(3).to(DEV), True)   # K % 8


def test_rcnn_heads_one_launch_per_layer_vs_rocblas(run):
    """the engine's RCNN cls / reg heads through jm_linear_rows vs the GEMM + bias + ReLU route"""
    eng = run["eng"]
    pts = run["inter"]["pts_input"]
    from jmodt_amd.profile import prof
    with torch.no_grad():
        prof.reset()
        prof.enabled = True
        try:
            a = eng.rcnn_forward(pts)
            torch.cuda.synchronize()
            names = set(prof.records)
        finally:
            prof.enabled = False
            prof.reset()
        eng.fuse_small_heads = False
        try:
            b = eng.rcnn_forward(pts)
        finally:
            eng.fuse_small_heads = True
        eng.fuse_head_stacks = True                     # (opt-in: measured slower than a launch per layer, detector.py)
        prof.reset()
        prof.enabled = True
        try:
            c = eng.rcnn_forward(pts)
            torch.cuda.synchronize()
            names_stack = set(prof.records)
        finally:
            prof.enabled = False
            prof.reset()
            eng.fuse_head_stacks = False
    for k in ("rcnn_cls", "rcnn_reg"):
        close(a[k], b[k])
        close(c[k], b[k])
        assert a[k].shape == b[k].shape == c[k].shape and a[k].is_contiguous()
    # one conv1d_stack launch per head where the RoI count is a multiple of 32, else one jm_linear_rows launch per layer
    R = a["rcnn_cls"].shape[0]
    assert any("linear_rows" in n for n in names), names
    if R % 32 == 0:                                     # one conv1d_stack launch per head instead
        assert not any("linear_rows" in n for n in names_stack), names_stack


@pytest.mark.parametrize("B,n,c0,c1,xyz1,widths,relus", [
    (2, 4096, 128, 0, False, [128, 77], [True, False]),          # both RPN heads as one block-diagonal stack
    (3, 1024, 96, 3, True, [64], [False]),                       # hoisted first SA layer: W_f f + W_x xyz^T
    (2, 256, 512, 256, False, [512, 512], [True, True]),         # an FP module: cat[interpolated, skip] -> SharedMLP
    (1, 64, 20, 7, False, [40, 24, 9], [True, True, False]),     # ragged widths, both operands staged, 3 layers
    (2, 32, 16, 0, False, [5], [True]),
    (1, 96, 259, 0, False, [300, 130], [True, True]),
])
@pytest.mark.parametrize("tile64", [0, 2])
def test_conv1d_stack_vs_fp64(B, n, c0, c1, xyz1, widths, relus, tile64, monkeypatch):
    """csrc/conv1d_stack.hip (32-point tiles) and csrc/conv1d_stack64.hip (64-point tiles wherever they fit) vs the same chain in
    fp64 torch"""
    import jmodt_amd.ops.conv1d as C1
    from jmodt_amd.ops.conv1d import PackedConv1dStack
    monkeypatch.setattr(C1, "TILE64", tile64)
    g = torch.Generator().manual_seed(B * 1000 + n + c0)
    cin = c0 + c1
    layers, k = [], cin
    for w, r in zip(widths, relus):
        layers.append(((torch.randn(w, k, generator=g) * (2.0 / k) ** 0.5).to(DEV), (torch.randn(w, generator=g) * 0.2).to(DEV), r))
        k = w
    x0 = torch.randn(B, c0, n, generator=g).to(DEV)
    x1 = (torch.randn(B, n, 3, generator=g) if xyz1 else torch.randn(B, c1, n, generator=g)).to(DEV) if c1 else None
    st = PackedConv1dStack(layers, c0, c1, xyz1)
    assert st.supported(B, n)
    got = st(x0, x1)
    h = x0.double() if x1 is None else torch.cat([x0.double(), (x1.transpose(1, 2) if xyz1 else x1).double()], dim=1)
    for W, b, r in layers:
        h = torch.einsum("oc,bcn->bon", W.double(), h) + b.double()[None, :, None]
        if r:
            h = torch.relu(h)
    assert got.shape == h.shape
    close(got, h)
    got_pm = st(x0, x1, point_major=True)
    assert got_pm.shape == (B, n, widths[-1]) and torch.equal(got_pm.transpose(1, 2), got)
    assert not PackedConv1dStack(layers, c0, c1, xyz1).supported(B, n + 1)          # n % 32


@pytest.mark.parametrize("B,H,W,cout", [(2, 48, 160, 64), (1, 5, 7, 16), (1, 33, 300, 64), (2, 8, 256, 4), (1, 384, 1280, 64)])
def test_conv3x3_rgb_bias_relu_vs_torch(B, H, W, cout):
    """csrc/conv_rgb.hip (3-channel 3x3 convolution + bias + ReLU, channels-last output) vs fp64 torch conv2d"""
    from jmodt_amd.ops.fusion import conv3x3_rgb_bias_relu
    g = torch.Generator().manual_seed(H * W + cout)
    img = torch.rand(B, 3, H, W, generator=g).to(DEV)
    Wt = (torch.randn(cout, 3, 3, 3, generator=g) * 0.3).to(DEV)
    b = (torch.randn(cout, generator=g) * 0.2).to(DEV)
    got = conv3x3_rgb_bias_relu(img, Wt, b)
    want = torch.relu(F.conv2d(img.double(), Wt.double(), b.double(), padding=1))
    assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
    assert (want > 0).any() and (want == 0).any()
    close(got, want)


@pytest.mark.parametrize("B,cin,cout,H,W,bias,relu", [
    (2, 64, 128, 48, 160, True, True), (1, 16, 64, 5, 7, True, True), (1, 32, 64, 33, 31, False, True), (2, 128, 256, 24, 80, True, False),
    (1, 256, 512, 12, 40, True, True), (3, 48, 192, 9, 18, True, True), (1, 64, 128, 192, 640, True, True)])
def test_conv3x3_wino_bias_relu_vs_torch(B, cin, cout, H, W, bias, relu):
    """csrc/conv_wino.hip (fused Winograd F(2x2, 3x3) + bias + ReLU, channels-last, fp32) vs fp64 torch conv2d: odd sizes (partial
    tiles and patches), one-chunk K, no bias / no ReLU, the three image-branch widths"""
    from jmodt_amd.ops.fusion import conv3x3_wino_bias_relu, pack_wino_weight, wino_supported
    assert wino_supported(cin, cout) and not wino_supported(cin + 4, cout) and not wino_supported(cin, cout + 16)
    g = torch.Generator().manual_seed(H * W + cout + cin)
    x = (torch.randn(B, cin, H, W, generator=g) + 0.2).to(DEV).contiguous(memory_format=torch.channels_last)
    Wt = (torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5).to(DEV)
    b = (torch.randn(cout, generator=g) * 0.2).to(DEV) if bias else None
    got = conv3x3_wino_bias_relu(x, pack_wino_weight(Wt), b, cout, relu=relu)
    want = F.conv2d(x.double(), Wt.double(), b.double() if bias else None, padding=1)
    want = torch.relu(want) if relu else want
    assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
    assert (want > 0).any() and ((want == 0).any() if relu else (want < 0).any())
    close(got, want)
    assert (got.double() - want).abs().max().item() <= 2e-6 * want.abs().max().item()      # what the direct fp32 form achieves too


def test_image_branch_with_and_without_the_winograd_kernels():
    """the engine's image pyramid at the benchmarked width (64 / 128 / 256 / 512 channels, BatchNorm folded): stride-1 convolutions on
    conv_wino.hip vs the same blocks on MIOpen + bias/ReLU pass, and vs the module's own eval-mode forward in float64"""
    from jmodt_amd.detector import DetectorConfig
    eng = make_engine(3, DetectorConfig.survey()).to(DEV).eval()
    img = torch.rand(2, 3, 96, 160, generator=torch.Generator().manual_seed(4)).to(DEV)

    def pyramid():
        cur, outs = img, []
        with torch.no_grad():
            for i in range(4):
                cur = eng._image_block(i, cur)
                outs.append(cur)
        return outs
    eng.wino_conv = True
    assert eng._wino_ok(eng.rpn.backbone_net.Img_Block[1].conv1, pyramid()[0])          # the kernel really is the one that runs
    got = pyramid()
    assert any(k.endswith(".wino") for k in eng._folded)
    eng.wino_conv = False
    eng.invalidate()
    lib = pyramid()
    assert not any(k.endswith(".wino") for k in eng._folded)
    blocks64 = [b.double() for b in eng.rpn.backbone_net.Img_Block]
    cur, want = img.double(), []
    with torch.no_grad():
        for b in blocks64:
            cur = b(cur)
            want.append(cur)
    eng.float()
    for g, l, w in zip(got, lib, want):
        close(g, w)
        close(l, w)
        close(g, l)


def test_engine_stream_safety_soak(run):
    """the same batch through 12 steps with every overlap / prefetch on and allocator churn on the main stream in
    between: the pyramids are handed between streams without record_stream (ordered release), so a recycling race would
    show up as a changed intermediate result"""
    eng = run["eng"]
    a, img, xy = T(run["xyz"]), T(run["img"]), T(run["xy"])
    g = torch.Generator().manual_seed(5)
    for i in range(12):
        with torch.no_grad():
            cache, aff, inter = eng(a, img, xy, next_xyz=a, next_image=img if i % 3 == 0 else None)
        junk = [torch.empty(int(s), device=DEV).fill_(float(i)) for s in torch.randint(1 << 8, 1 << 20, (5,), generator=g).tolist()]
        del junk
        for k in ("backbone_features", "rois", "rcnn_feat", "pred_boxes3d"):
            assert torch.equal(inter[k], run["inter"][k]), (i, k)
        assert torch.equal(cache.count, run["cache"].count)
    with torch.no_grad():
        eng(a, img, xy)                                    # consume the last announcement
    assert eng._prefetched == [] and eng._prefetched_img is None


# ------------------------------------------------------------------------------------------------------------------
# the reference's COMPLETE forward (tests/golden/forward_ref.npz: PointRCNN.forward in TEST mode, executed in the
# authoring container over the CPU oracle's extension entry points, tests/golden/make_golden_forward.py)
# ------------------------------------------------------------------------------------------------------------------
def test_engine_matches_the_references_complete_forward():
    """DetectAffinityEngine with the SAME weights (same parameter names: strict load) on the same frames: backbone + LI-Fusion +
    RPN heads free running; proposal layer, RoI pooling + canonical transform and the RCNN teacher-forced on the reference's own
    intermediate outputs; fused and un-fused, with and without the duplicate compaction"""
    from jmodt_amd.detector import DetectAffinityEngine
    from tests.test_oracle_cpu import reference_forward_fixture
    cfg, sd, g = reference_forward_fixture()
    eng = DetectAffinityEngine(cfg)
    own = eng.state_dict()
    extra = {k: v for k, v in own.items() if k not in sd}                      # (the engine's BatchNorm counters)
    assert all(k.endswith("num_batches_tracked") for k in extra) and not [k for k in sd if k not in own]
    eng.load_state_dict({**extra, **sd}, strict=True)
    eng = eng.to(DEV).eval()
    xyz, img, xy = T(g["xyz"]), T(g["img"]), T(g["pts_xy"])
    ref_rpn = dict(backbone_xyz=xyz, backbone_features=T(g["out.backbone_features"]), rpn_cls=T(g["out.rpn_cls"]), rpn_reg=T(g["out.rpn_reg"]))
    for fuse, dedupe in ((True, True), (True, False), (False, False)):
        for m_ in eng.modules():
            if hasattr(m_, "fuse"):
                m_.fuse = fuse
        eng.dedupe_rcnn = dedupe
        with torch.no_grad():
            cache, aff, inter = eng(xyz, img, xy)
            close(inter["backbone_features"], g["out.backbone_features"])
            close(inter["rpn_cls"], g["out.rpn_cls"]); close(inter["rpn_reg"], g["out.rpn_reg"])
            rois, scores = eng.proposals(ref_rpn)
            close(rois, g["out.rois"]); close(scores, g["out.roi_scores_raw"], 1e-6)
            pts = eng.roi_pool(ref_rpn, T(g["out.rois"]))
            got = pts.cpu().numpy()
            assert np.array_equal(got[..., 3], g["out.pts_input_geom"][..., 3])
            close(got[..., :3], g["out.pts_input_geom"][..., :3]); close(got[..., 4], g["out.pts_input_geom"][..., 4], 1e-6)
            close(got.astype(np.float64).sum(axis=(1, 2)), g["out.pts_input_sum"], 1e-6)
            out = eng.rcnn_forward(pts)
            close(out["rcnn_feat"], g["out.rcnn_feat"]); close(out["rcnn_cls"], g["out.rcnn_cls"]); close(out["rcnn_reg"], g["out.rcnn_reg"])
    assert 0 < g["out.pts_input_geom"][..., 3].mean() < 1 and np.abs(g["out.rcnn_feat"]).max() > 1.0
