"""Generate the committed golden vectors (tests/golden/*.npz).

Run in the authoring container, where /root/reference exists:
    python tests/golden/make_golden.py

Two kinds of vectors (each file records which, in its `source` field):
  * "reference": expected outputs come from the REFERENCE ITSELF executed here on CPU —
      - roipool3d: the reference's own roipool3d.cpp CPU functions (oracle/_ref, compiled from
        /root/reference/jmodt/ops/roipool3d/src/roipool3d.cpp);
      - affinity: the reference's layer builder jmodt/ops/pointnet2/pytorch_utils.py (imported
        from /root/reference) assembled as rcnn.py:91-111 does, evaluated with the torch ops of
        tracker.py:81-112;  jmodt.config is NOT imported (it needs `easydict`, absent here), the
        three config values it would supply are written out below with their file:line;
      - boxes3d_to_bev / enlarge_box3d: jmodt/utils/kitti_utils.py imported from /root/reference;
      - feature_gather: torch.nn.functional.grid_sample, the op the reference calls.
  * "oracle": the reference has no CPU code for the op (FPS, ball_query, group, three_nn,
    interpolate, BEV overlap, NMS); expected outputs come from oracle/jmodt_oracle.c, which
    tests/test_oracle_cpu.py pins against independent numpy restatements.  These files pin the
    oracle against regressions and give the GPU tests fixed byte-level targets.
Only data is stored: seeded inputs and expected outputs.  No reference source text.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from jmodt_amd import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402

REFERENCE = "/root/reference"


def save(name, **kw):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **kw)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def gen_oracle_vectors():
    # FPS: random floats, duplicates (ties), quantised grid (contraction-proof), non-power-of-two n
    cases = {}
    for tag, B, N, m, kw in [("rand", 2, 1024, 256, {}), ("dup", 1, 2048, 256, dict(dup_frac=0.3)),
                             ("grid", 1, 1024, 128, dict(quantize=2.0 ** -3)), ("n1000", 2, 1000, 128, {}),
                             ("n16384", 1, 16384, 1024, dict(dup_frac=0.1))]:
        xyz = synth.cloud(B, N, seed=101, **kw)
        cases[f"{tag}_xyz"] = xyz
        cases[f"{tag}_idx"] = orc.furthest_point_sample(xyz, m)
    save("fps.npz", source="oracle", **cases)

    cases = {}
    xyz = synth.cloud(2, 2048, 202, dup_frac=0.1)
    new_xyz = xyz[:, ::8].copy()
    dense = synth.dense_cloud(2, 1024, 203)
    dnew = dense[:, ::8].copy()
    cases.update(xyz=xyz, new_xyz=new_xyz, dense=dense, dnew=dnew)
    for r, ns in [(0.1, 16), (0.5, 32), (4.0, 64)]:
        cases[f"sparse_r{r}_ns{ns}"] = orc.ball_query(r, ns, xyz, new_xyz)
    for r, ns in [(0.4, 16), (1.0, 32), (2.0, 64)]:
        cases[f"dense_r{r}_ns{ns}"] = orc.ball_query(r, ns, dense, dnew)
    save("ball_query.npz", source="oracle", **cases)

    rng = np.random.default_rng(303)
    unknown = synth.cloud(2, 512, 304, dup_frac=0.1)
    known = unknown[:, ::4].copy()
    d2, idx = orc.three_nn(unknown, known)
    feats = rng.normal(size=(2, 8, known.shape[1])).astype(np.float32)
    w = 1.0 / (np.sqrt(d2) + 1e-8)
    w = (w / w.sum(2, keepdims=True)).astype(np.float32)
    save("three_nn_interp.npz", source="oracle", unknown=unknown, known=known, dist2=d2, idx=idx, feats=feats,
         weight=w, out=orc.three_interpolate(feats, idx, w))

    boxes, scores = synth.bev_boxes(1000, 404)
    out = dict(boxes=boxes, scores=scores)
    for thr in (0.1, 0.8, 0.85):
        out[f"normal_{thr}"] = orc.nms(boxes, scores, thr, normal=True)
    b2, s2 = synth.bev_boxes(300, 405)
    out.update(boxes_rot=b2, scores_rot=s2)
    for thr in (0.1, 0.5):
        out[f"rot_{thr}"] = orc.nms(b2, s2, thr, normal=False)
    a, _ = synth.bev_boxes(40, 406, extent=6.0)
    b, _ = synth.bev_boxes(30, 407, extent=6.0)
    out.update(pair_a=a, pair_b=b, overlap=orc.boxes_overlap_bev(a, b), iou=orc.boxes_iou_bev(a, b))
    save("iou3d_nms.npz", source="oracle", **out)


def gen_reference_vectors():
    import torch
    import torch.nn.functional as F
    if not os.path.isdir(os.path.join(REFERENCE, "jmodt")):
        raise SystemExit("needs /root/reference")
    sys.path.insert(0, REFERENCE)

    # ---- roipool3d: the reference's own CPU functions (oracle/_ref)
    sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref"))
    import roipool3d_ref
    from jmodt.utils import kitti_utils
    pts = synth.dense_cloud(2, 4096, 501, extent=14.0)
    pts[..., 1] = pts[..., 1] / 7.0
    boxes = synth.proposals(pts, 24, 502)
    boxes[0, 0, 0:3] = [500, 0, 500]          # empty
    boxes[0, 1, 3:6] = [60, 60, 60]           # huge: 10 m cut-off and > S points
    boxes[1, 2, 3:6] = [1.0, 0.8, 1.5]        # small: 0 < count < S -> cyclic padding
    feat = np.random.default_rng(503).normal(size=(2, 4096, 6)).astype(np.float32)
    S = 128
    enlarged = np.stack([kitti_utils.enlarge_box3d(boxes[b], 0.2) for b in range(2)])
    pooled = np.zeros((2, 24, S, 9), np.float32)
    empty = np.zeros((2, 24), np.int32)
    flags = np.zeros((2, 24, 4096), np.int64)
    for b in range(2):
        tp, tb, tf = torch.from_numpy(pts[b]), torch.from_numpy(enlarged[b]), torch.from_numpy(feat[b])
        fl = torch.zeros((24, 4096), dtype=torch.int64)
        roipool3d_ref.pts_in_boxes3d_cpu(fl, tp, tb)
        pp, pf, ef = torch.zeros((24, S, 3)), torch.zeros((24, S, 6)), torch.zeros((24,), dtype=torch.int64)
        roipool3d_ref.roipool3d_cpu(tp, tb, tf, pp, pf, ef)
        pooled[b, :, :, :3], pooled[b, :, :, 3:] = pp.numpy(), pf.numpy()
        empty[b], flags[b] = ef.numpy().astype(np.int32), fl.numpy()
    save("roipool3d_ref.npz", source="reference", pts=pts, feat=feat, boxes=boxes, enlarged=enlarged,
         pooled=pooled, empty=empty, flags=flags.astype(np.uint8), S=S)

    # ---- boxes3d_to_bev (reference kitti_utils)
    b3 = synth.proposals(synth.dense_cloud(1, 128, 9, extent=30.0), 50, 10)[0]
    save("kitti_utils_ref.npz", source="reference", boxes3d=b3,
         bev=kitti_utils.boxes3d_to_bev_torch(torch.from_numpy(b3)).numpy(),
         enlarged=kitti_utils.enlarge_box3d(b3, 0.2))

    # ---- affinity: reference layer builder + torch ops
    from jmodt.ops.pointnet2 import pytorch_utils as ref_pt  # the reference's own module
    LINK_FC = SE_FC = [512, 512]   # jmodt/config.py:166-169 (REID.LINK_FC / SE_FC)
    DP_RATIO = 0.0                 # jmodt/config.py RCNN.DP_RATIO (rcnn.py:98-99 reads RCNN's ratio)
    USE_BN = False                 # jmodt/config.py REID.USE_BN

    def build(seed):
        torch.manual_seed(seed)
        layers, pre = [], 512
        for k in LINK_FC:
            layers.append(ref_pt.Conv1d(pre, k, bn=USE_BN)); pre = k
        layers.append(ref_pt.Conv1d(pre, 1, activation=None))
        layers.insert(1, torch.nn.Dropout(DP_RATIO))
        head = torch.nn.Sequential(*layers)
        for mod in head.modules():                      # rcnn.py:116-134 init_weights('xavier')
            if isinstance(mod, torch.nn.Conv1d):
                torch.nn.init.xavier_normal_(mod.weight)
                torch.nn.init.normal_(mod.bias, 0, 0.05)  # non-zero so the bias path is exercised
        return head.eval()

    link, se = build(0), build(1)
    out = {}
    for name, mod in (("link", link), ("se", se)):
        for k, v in mod.state_dict().items():
            out[f"{name}.{k}"] = v.numpy()
    for tag, P, D in (("64x64", 64, 64), ("3x5", 3, 5), ("1x1", 1, 1)):
        pf, df = synth.roi_features(P, 512, 600 + P), synth.roi_features(D, 512, 700 + D)
        tp, td = torch.from_numpy(pf), torch.from_numpy(df)
        with torch.no_grad():                            # tracker.py:81-112
            cor = torch.abs(tp.unsqueeze(1).repeat(1, D, 1) - td.unsqueeze(0).repeat(P, 1, 1))
            s = link(cor.view(P * D, -1, 1)).view(P, D)
            A = (torch.softmax(s, dim=1) + torch.softmax(s, dim=0)) / 2
            start = se(cor.mean(dim=0).unsqueeze(-1)).flatten()
            end = se(cor.mean(dim=1).unsqueeze(-1)).flatten()
        out.update({f"{tag}_pf": pf, f"{tag}_df": df, f"{tag}_raw": s.numpy(), f"{tag}_A": A.numpy(),
                    f"{tag}_start": start.numpy(), f"{tag}_end": end.numpy()})
    save("affinity_ref.npz", source="reference", **out)

    # ---- feature_gather: torch grid_sample (the op the reference calls)
    rng = np.random.default_rng(800)
    fm = rng.normal(size=(2, 8, 24, 80)).astype(np.float32)
    xy = rng.uniform(-1.1, 1.1, (2, 300, 2)).astype(np.float32)
    xy[0, 0] = [-1, -1]; xy[0, 1] = [1, 1]; xy[0, 2] = [1.5, 0.2]
    xy[0, 3] = [2 * 5 / 79 - 1, 2 * 7 / 23 - 1]
    want = F.grid_sample(torch.from_numpy(fm), torch.from_numpy(xy).unsqueeze(1), align_corners=True).squeeze(2)
    save("feature_gather_ref.npz", source="reference", fmap=fm, xy=xy, out=want.numpy())

    # ---- geometry helpers of the reference that the "next" rows build on (kitti_utils.py)
    # canonical transformation: proposal_target_layer.py:106-112 = subtract the RoI centre, then
    # rotate_pc_along_y_torch (kitti_utils.py:46-64) — executed with the reference's function
    rng = np.random.default_rng(21)
    rois = synth.proposals(synth.dense_cloud(1, 256, 22, extent=25.0), 12, 23)[0]                   # (12, 7)
    pooled_xyz = (rois[:, None, 0:3] + rng.normal(0, 1.5, (12, 40, 3))).astype(np.float32)           # points near each RoI
    canon = torch.from_numpy(pooled_xyz.copy())
    canon -= torch.from_numpy(rois[:, 0:3]).unsqueeze(1)
    canon = kitti_utils.rotate_pc_along_y_torch(canon, torch.from_numpy(rois[:, 6]))
    # tracker distance term: data_association.py:10-28 written with the reference's (numpy) corner function
    # boxes3d_to_corners3d (kitti_utils.py:66-104; its torch twin :107-133 allocates torch.cuda tensors and
    # cannot run here): 1 - centre distance / largest of the 64 corner-pair distances
    ba = synth.proposals(synth.dense_cloud(1, 256, 24, extent=25.0), 9, 25)[0]
    bb = synth.proposals(synth.dense_cloud(1, 256, 26, extent=25.0), 7, 27)[0]
    ca, cb = kitti_utils.boxes3d_to_corners3d(ba), kitti_utils.boxes3d_to_corners3d(bb)              # (M, 8, 3)
    centre = np.linalg.norm(ba[:, None, :3] - bb[None, :, :3], axis=-1)
    corner = np.linalg.norm(ca[:, None, :, None, :] - cb[None, :, None, :, :], axis=-1).reshape(9, 7, 64).max(-1)
    save("geometry_ref.npz", source="reference", rois=rois, pooled_xyz=pooled_xyz, canonical=canon.numpy(),
         boxes_a=ba, boxes_b=bb, corners_a=ca, boxes_dist=(1.0 - centre / corner).astype(np.float32))

    # ---- layer builders: the reference's SharedMLP / Conv1d / FC (pytorch_utils.py) in eval mode with random
    # BatchNorm statistics; state_dict + input + output.  Pins the build's mirror of the builders (same
    # parameter names, same arithmetic) that the SA / FP modules and the fused SA kernel's BN folding rest on.
    torch.manual_seed(11)
    g = torch.Generator().manual_seed(12)
    mlp = ref_pt.SharedMLP([9, 16, 24, 40], bn=True)
    conv1 = ref_pt.Conv1d(12, 20, bn=True)
    fc = ref_pt.FC(10, 6, bn=True)
    for mod in (mlp, conv1, fc):
        for m_ in mod.modules():
            if isinstance(m_, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
                with torch.no_grad():
                    m_.weight.copy_(torch.rand(m_.weight.shape, generator=g) + 0.5)
                    m_.bias.copy_(torch.randn(m_.bias.shape, generator=g) * 0.2)
                    m_.running_mean.copy_(torch.randn(m_.running_mean.shape, generator=g) * 0.3)
                    m_.running_var.copy_(torch.rand(m_.running_var.shape, generator=g) + 0.5)
        mod.eval()
    x_mlp = torch.randn(2, 9, 7, 5, generator=g)
    x_c1 = torch.randn(3, 12, 11, generator=g)
    x_fc = torch.randn(4, 10, generator=g)
    out = {"source": "reference", "x_mlp": x_mlp.numpy(), "x_conv1d": x_c1.numpy(), "x_fc": x_fc.numpy()}
    with torch.no_grad():
        out["y_mlp"], out["y_conv1d"], out["y_fc"] = mlp(x_mlp).numpy(), conv1(x_c1).numpy(), fc(x_fc).numpy()
    for tag, mod in (("mlp", mlp), ("conv1d", conv1), ("fc", fc)):
        for k, v in mod.state_dict().items():
            out[f"{tag}.{k}"] = v.numpy()
    save("layer_builders_ref.npz", **out)


if __name__ == "__main__":
    gen_oracle_vectors()
    gen_reference_vectors()
