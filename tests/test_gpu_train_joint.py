"""GPU tier (-m gpu): the joint-mode training step (jmodt_amd/train_joint.py; tools/train.py:96-107 without FINETUNE, BASELINE
configs[3]'s 66.9 MB gradient): the differentiable composition of the detector is the SAME network as the fused inference
engine, and one step sends a finite gradient into every parameter."""
import numpy as np
import pytest
import torch

from jmodt_amd import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def close(got, want, tol=1e-4):
    got, want = got.detach().double().cpu(), want.detach().double().cpu()
    scale = max(1.0, float(want.abs().max()))
    assert float((got - want).abs().max()) <= tol * scale, (float((got - want).abs().max()), scale)


@pytest.fixture(scope="module")
def tiny():
    from jmodt_amd.detector import DetectorConfig
    from tests.test_gpu_detector import make_engine
    eng = make_engine(seed=3, cfg=DetectorConfig.tiny()).to(DEV).eval()
    xyz, img, xy = synth.frames(2, 2048, 77, H=96, W=320, native=(94, 310))
    # (synth.frames projects with the KITTI intrinsics of the 1280-wide canvas: on this 320-wide smoke image almost every point
    # falls outside and the LI-Fusion gather returns its zero padding; here the image side must carry signal AND gradient)
    xy = np.random.default_rng(5).uniform(-0.98, 0.98, size=xy.shape).astype(np.float32)
    return eng, T(xyz), T(img), T(xy)


def test_joint_forward_is_the_inference_engines_network(tiny):
    """eval-mode BatchNorm on both sides: the differentiable route (un-fused operators, module forwards) against the fused,
    folded kernels of DetectAffinityEngine — backbone features and RPN heads free running, the RCNN on the same pooled points"""
    from jmodt_amd.train_joint import joint_forward, rcnn_forward_train
    eng, xyz, img, xy = tiny
    with torch.no_grad():
        want = eng.rpn_forward(xyz, img, xy)
    with torch.enable_grad():
        got = joint_forward(eng, xyz, img, xy, rois_per_frame=eng.cfg.rpn_post_nms_top_n)
    assert got["rpn_cls"].requires_grad and got["rcnn_reg"].requires_grad and got["rcnn_feat"].requires_grad
    close(got["backbone_features"], want["backbone_features"])
    close(got["rpn_cls"], want["rpn_cls"])
    close(got["rpn_reg"], want["rpn_reg"])
    with torch.no_grad():
        rois, _ = eng.proposals(want)
        pts_input = eng.roi_pool(want, rois)
        ref = eng.rcnn_forward(pts_input)
    with torch.enable_grad():
        mine = rcnn_forward_train(eng.rcnn_net, pts_input)
    close(mine["rcnn_cls"], ref["rcnn_cls"])
    close(mine["rcnn_reg"], ref["rcnn_reg"])
    close(mine["rcnn_feat"], ref["rcnn_feat"].squeeze(-1))


def test_joint_step_reaches_every_parameter_and_updates_it(tiny):
    """one step in train mode: every one of the detector's and the affinity heads' parameters receives a finite gradient, almost
    all of them non-zero, and the fused Adam moves them; no process group: no collective is issued"""
    from jmodt_amd import train_joint
    from jmodt_amd.detector import DetectorConfig
    from tests.test_gpu_detector import make_engine
    _, xyz, img, xy = tiny
    eng = make_engine(seed=3, cfg=DetectorConfig.tiny()).to(DEV).train()      # (a fresh one: the step changes the weights)
    for p in eng.parameters():
        p.requires_grad_(True)
    params = list(eng.parameters())
    opt = torch.optim.Adam(params, lr=1e-3, fused=True)
    R = min(64, eng.cfg.rpn_post_nms_top_n)
    tids = torch.randint(0, 6, (2, R), generator=torch.Generator().manual_seed(4)).float().to(DEV)
    before = [p.detach().clone() for p in params]
    loss = train_joint.joint_step(eng, xyz, img, xy, tids, opt, world=1, rois_per_frame=R)
    torch.cuda.synchronize()
    assert torch.isfinite(loss) and train_joint.LAST_GRAD_COLLECTIVES == 0
    missing = [n for n, p in eng.named_parameters() if p.grad is None]
    assert not missing, missing
    assert all(bool(torch.isfinite(p.grad).all()) for p in params)
    # the only parameters with an all-zero gradient are the biases of convolutions that feed a TRAIN-mode BatchNorm (the batch
    # mean removes them: d loss / d bias = 0 exactly) — the fusion convolutions of backbone.py:44-81 and image_fusion_conv
    zero = sorted(n for n, p in eng.named_parameters() if not bool((p.grad != 0).any()))
    expect = sorted([f"rpn.backbone_net.Fusion_Conv.{i}.{m}.bias" for i in range(len(eng.rpn.backbone_net.Fusion_Conv))
                     for m in ("conv1", "IA_Layer.conv1.0")] +
                    ["rpn.backbone_net.final_fusion_img_point.conv1.bias", "rpn.backbone_net.final_fusion_img_point.IA_Layer.conv1.0.bias",
                     "rpn.backbone_net.image_fusion_conv.bias"])
    assert set(zero) <= set(expect), sorted(set(zero) - set(expect))
    # (their gradient is zero up to rounding: some come out as 1e-12 instead of 0 and do not move a float32 weight either)
    stuck = [n for (n, p), a in zip(eng.named_parameters(), before) if torch.equal(a, p.detach()) and n not in expect]
    assert not stuck, stuck
    # the gradient that reaches the backbone comes from the RPN heads only: roipool3d is not differentiable (as in the reference)
    assert float(eng.rpn.backbone_net.SA_modules[0].mlps[0][0].conv.weight.grad.abs().max()) > 0
    assert float(eng.rpn.backbone_net.Img_Block[0].conv1.weight.grad.abs().max()) > 0
    assert float(eng.rcnn_net.xyz_up_layer[0].conv.weight.grad.abs().max()) > 0
    assert float(eng.rcnn_net.link_layer[0].conv.weight.grad.abs().max()) > 0


def test_bench_joint_training_step_under_the_launcher():
    """bench.py --workload train --joint on the launcher path (one-rank RCCL group): the bucketed all-reduce of ALL parameters
    is issued and timed, bytes = 4 x the parameter count"""
    import json
    import os
    import subprocess
    import sys
    from jmodt_amd.detector import DetectAffinityEngine, DetectorConfig
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
           "--workload", "train", "--joint", "--tiny", "--launch", "--batch", "2"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and len(lines[0]) <= 4096, p.stdout[-2000:]
    r = json.loads(lines[0])
    nparams = sum(q.numel() for q in DetectAffinityEngine(DetectorConfig.tiny()).parameters())
    ga = r["grad_allreduce"]
    assert ga["issued"] >= 1 and ga["bytes_per_step"] == 4 * nparams and ga["ms_per_step"] > 0 and r["value"] > 0
    assert "joint" in r["config"]["workload"]


def _route_forward(route, eng, xyz, img, xy):
    """(backbone features (B, C, N), rpn_cls (B, N, 1), rpn_reg (B, N, C), rcnn(pts) -> dict) of one training route on the module
    containers of `eng`: "operators" = train_joint.backbone_forward / rcnn_forward_train (torch autograd over the grouped tensors),
    "rows" = train_rows.rpn_forward_rows / rcnn_forward_rows (csrc/rows_*.hip forward and backward, BatchNorm folded) — what
    joint_step(route="auto") and rcnn_step run"""
    if route == "operators":
        from jmodt_amd.train_joint import backbone_forward, rcnn_forward_train
        feats = backbone_forward(eng.rpn.backbone_net, xyz, img, xy)
        rpn_cls = eng.rpn.rpn_cls_layer(feats).transpose(1, 2)
        rpn_reg = eng.rpn.rpn_reg_layer(feats).transpose(1, 2)
        return feats, rpn_cls, rpn_reg, lambda pts: rcnn_forward_train(eng.rcnn_net, pts)
    from jmodt_amd.train_rows import BnFold, rcnn_forward_rows, rpn_forward_rows
    fold = BnFold(eng)
    out = rpn_forward_rows(eng, xyz, img, xy, fold)
    # (a fold object serves ONE backward: the RCNN half of the loss is back-propagated on its own, with its own fold)
    return out["backbone_features"], out["rpn_cls"], out["rpn_reg"], lambda pts: rcnn_forward_rows(eng, pts, BnFold(eng), None)


@pytest.mark.parametrize("route", ["operators", "rows"])
def test_joint_forward_matches_the_references_complete_forward(route):
    """the DIFFERENTIABLE composition (train_joint.joint_forward / rcnn_forward_train) against the reference's own
    `PointRCNN.forward` executed over the oracle's extension entry points (tests/golden/forward_ref.npz, the fixture the inference
    engine is checked against): same weights by name, eval-mode BatchNorm, backbone + RPN heads free running, the RCNN on the
    pooled points the engine forms from the REFERENCE's proposals.  With this the training route of round 4 is pinned to the
    reference's Python as well, not only to the engine"""
    from jmodt_amd.detector import DetectAffinityEngine
    from tests.test_gpu_detector import close as close_np
    from tests.test_oracle_cpu import reference_forward_fixture
    cfg, sd, g = reference_forward_fixture()
    eng = DetectAffinityEngine(cfg)
    own = eng.state_dict()
    eng.load_state_dict({**{k: v for k, v in own.items() if k not in sd}, **sd}, strict=True)
    eng = eng.to(DEV).eval()
    xyz, img, xy = T(g["xyz"]), T(g["img"]), T(g["pts_xy"])
    for p in eng.parameters():
        p.requires_grad_(True)
    with torch.enable_grad():
        feats, rpn_cls, rpn_reg, rcnn = _route_forward(route, eng, xyz, img, xy)
    assert feats.requires_grad
    close_np(feats, g["out.backbone_features"])
    close_np(rpn_cls, g["out.rpn_cls"])
    close_np(rpn_reg, g["out.rpn_reg"])
    ref_rpn = dict(backbone_xyz=xyz, backbone_features=T(g["out.backbone_features"]), rpn_cls=T(g["out.rpn_cls"]), rpn_reg=T(g["out.rpn_reg"]))
    with torch.no_grad():
        pts = eng.roi_pool(ref_rpn, T(g["out.rois"]))
    with torch.enable_grad():
        out = rcnn(pts)
    close_np(out["rcnn_feat"].unsqueeze(-1), g["out.rcnn_feat"])
    close_np(out["rcnn_cls"], g["out.rcnn_cls"])
    close_np(out["rcnn_reg"], g["out.rcnn_reg"])


@pytest.mark.parametrize("route", ["operators", "rows"])
def test_joint_backward_matches_the_references_autograd(route):
    """gradients of ALL detector parameters through each training route — "operators": jm_*_grad kernels + torch autograd on the module
    containers; "rows": the hand-written forward / backward row kernels with folded BatchNorm, the DEFAULT of joint_step and what
    rcnn_step runs — against the reference's own backward — `model.rpn(input)` / `PointRCNN.forward` with autograd on, its
    pointnet2_utils Functions bound to the CPU oracle (tests/golden/make_golden_backward.py -> backward_ref.npz: per tensor the
    L2 norm, the sum, max |g| and 64 entries).  Same weights / frames as forward_ref.npz, eval-mode BatchNorm, the thin loss of
    train_joint without its re-id term.  240 tensors: backbone SA / FP / image blocks / LI-Fusion / deconvolutions, RPN heads,
    RCNN lift, set abstraction and heads"""
    import json
    import os
    from jmodt_amd.detector import DetectAffinityEngine
    from tests.test_oracle_cpu import reference_forward_fixture
    ref = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "backward_ref.npz"))
    names = json.loads(str(ref["names"]))
    cfg, sd, g = reference_forward_fixture()
    eng = DetectAffinityEngine(cfg)
    own = eng.state_dict()
    eng.load_state_dict({**{k: v for k, v in own.items() if k not in sd}, **sd}, strict=True)
    eng = eng.to(DEV).eval()
    for p in eng.parameters():
        p.requires_grad_(True)
    xyz, img, xy = T(g["xyz"]), T(g["img"]), T(g["pts_xy"])
    N = xyz.shape[1]
    with torch.enable_grad():
        _, rpn_cls, rpn_reg, rcnn = _route_forward(route, eng, xyz, img, xy)
        loss_rpn = (rpn_cls.sum() + rpn_reg.sum()) / N
    loss_rpn.backward()
    torch.cuda.synchronize()
    assert abs(loss_rpn.item() - float(ref["loss_rpn"])) < 1e-4 * max(1.0, abs(float(ref["loss_rpn"])))
    grads = {k: p.grad.detach().clone() for k, p in eng.named_parameters() if p.grad is not None}
    assert all(k.startswith("rpn.") for k in grads)
    eng.zero_grad(set_to_none=True)
    ref_rpn = dict(backbone_xyz=xyz, backbone_features=T(g["out.backbone_features"]), rpn_cls=T(g["out.rpn_cls"]), rpn_reg=T(g["out.rpn_reg"]))
    with torch.no_grad():
        pts = eng.roi_pool(ref_rpn, T(g["out.rois"]))
    with torch.enable_grad():
        out = rcnn(pts)
        loss_rcnn = out["rcnn_cls"].sum() + out["rcnn_reg"].sum()
    loss_rcnn.backward()
    torch.cuda.synchronize()
    assert abs(loss_rcnn.item() - float(ref["loss_rcnn"])) < 1e-4 * abs(float(ref["loss_rcnn"]))
    grads.update({k: p.grad.detach().clone() for k, p in eng.named_parameters() if p.grad is not None})
    assert sorted(grads) == names, (sorted(set(names) - set(grads))[:5], sorted(set(grads) - set(names))[:5])
    worst = (0.0, None)
    for i, k in enumerate(names):
        flat = grads[k].double().reshape(-1).cpu()
        norm, total, mx = ref["stats"][i]
        pos = torch.from_numpy(np.floor(np.linspace(0, flat.numel() - 1, ref["samples"].shape[1])).astype(np.int64))
        scale = max(float(mx), 1e-6)
        e_s = float((flat[pos] - torch.from_numpy(ref["samples"][i]).double()).abs().max()) / scale
        e_n = abs(float(flat.norm()) - norm) / max(norm, 1e-6)
        e_m = abs(float(flat.abs().max()) - mx) / scale
        worst = max(worst, (max(e_s, e_n, e_m), k))
        assert e_s <= 5e-4 and e_n <= 5e-4 and e_m <= 5e-4, (k, e_s, e_n, e_m, norm, mx)
    print("worst relative gradient error", worst)
    assert worst[0] > 0                                        # (not a comparison of a thing with itself)


@pytest.mark.parametrize("mode", ["joint", "rcnn"])
def test_joint_step_data_parallel_equals_single_process(tmp_path, mode):
    """mode "rcnn": the same for the RPN-fixed step (train_joint.rcnn_forward_backward; tools/train.py:104 with config.py:57) — only
    the RCNN's and the re-id heads' gradients exist and are exchanged.  mode "joint":
    the data-parallel joint step (tools/train.py:86-107: nn.DataParallel's scatter / gather / gradient reduction as one process
    per GPU): two ranks, each on its pair-aligned half of a 4-frame batch, re-id element counts and the gradients of all parameters
    all-reduced (SUM, several buckets) — against ONE process on the whole batch.  The ranks share cuda:0 and exchange over gloo
    (RCCL cannot put two ranks on one device); eval-mode BatchNorm, so that no statistic depends on the shard"""
    import os
    import socket
    import subprocess
    import sys
    helper = os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers", "joint_dp_worker.py")
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    extra = ["--rcnn"] if mode == "rcnn" else []
    procs = [subprocess.Popen([sys.executable, helper, str(tmp_path / f"r{r}.pt")] + extra, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              env=dict(base, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)))
             for r in range(2)]
    single = subprocess.run([sys.executable, helper, str(tmp_path / "one.pt"), "--single"] + extra, capture_output=True, text=True, timeout=900, env=base)
    assert single.returncode == 0, single.stderr[-3000:]
    for p in procs:
        _, err = p.communicate(timeout=900)
        assert p.returncode == 0, err[-3000:]
    one = torch.load(tmp_path / "one.pt")
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert r0["frames"] == (0, 2) and r1["frames"] == (2, 4) and one["collectives"] == 0 and r0["collectives"] >= (1 if mode == "rcnn" else 2)
    if mode == "rcnn":
        assert one["grads"] and all(k.startswith("rcnn_net.") for k in one["grads"])
    assert abs(r0["loss"] + r1["loss"] - one["loss"]) < 1e-4 * max(1.0, abs(one["loss"]))     # the shards' losses ADD
    assert sorted(r0["grads"]) == sorted(one["grads"])
    worst = 0.0
    for k, ref in one["grads"].items():
        assert torch.equal(r0["grads"][k], r1["grads"][k]), k                                 # replicas hold the same reduced gradient
        # (+ 1e-7 absolute: the link head's last bias has a gradient of exactly zero in exact arithmetic — the dual softmax is
        # invariant to a constant added to every score — and rounding residues of 1e-9 on either side)
        scale = float(ref.abs().max())
        err = float((r0["grads"][k] - ref).abs().max())
        worst = max(worst, err / max(scale, 1e-3))
        assert err <= 2e-4 * scale + 1e-7, (k, err, scale)
    print("worst relative gradient difference DP vs single process:", worst)


def test_rcnn_step_trains_the_rcnn_under_a_frozen_rpn(tiny):
    """the reference's default training mode (config.py:57 RPN.FIXED = True, point_rcnn.py:28-31, tools/train.py:104): the fused
    no-grad engine for the RPN half, the RCNN and the re-id heads on the row kernels.  Every RCNN tensor receives a finite gradient and
    moves; no RPN tensor has a gradient or changes; the step's RCNN outputs are the fused inference RCNN's on the same pooled points;
    after the optimizer step the inference engine runs on the UPDATED RCNN weights (its packed copies of them are re-made, the RPN's
    ~100 packed weights are not: detector._refresh's two signature groups)"""
    from jmodt_amd import train_joint
    from jmodt_amd.detector import DetectorConfig
    from tests.test_gpu_detector import make_engine
    _, xyz, img, xy = tiny
    eng = make_engine(seed=3, cfg=DetectorConfig.tiny()).to(DEV)
    opt = torch.optim.Adam(eng.rcnn_net.parameters(), lr=1e-3, fused=True)
    K = min(64, eng.cfg.rpn_post_nms_top_n)
    tids = torch.randint(0, 6, (2, K), generator=torch.Generator().manual_seed(4)).float().to(DEV)
    with pytest.raises(RuntimeError, match="prepare_rcnn"):
        train_joint.rcnn_step(eng, xyz, img, xy, tids, opt, rois_per_frame=K)
    train_joint.prepare_rcnn(eng)
    for m in eng.modules():                     # (Dropout off: the outputs are compared with the inference RCNN's)
        if isinstance(m, torch.nn.Dropout):
            m.eval()
    rcnn_names = [n for n, _ in eng.named_parameters() if n.startswith("rcnn_net.")]
    before = {n: p.detach().clone() for n, p in eng.named_parameters()}
    # forward + backward alone first: outputs against the fused inference RCNN on the same RoIs
    loss, out = train_joint.rcnn_forward_backward(eng, xyz, img, xy, tids, rois_per_frame=K)
    with torch.no_grad():
        rpn_out = eng.rpn_forward(xyz, img, xy)
        rois, _ = eng.proposals(rpn_out)
        assert torch.equal(rois[:, :K], out["rois"])
        ref = eng.rcnn_forward(eng.roi_pool(rpn_out, rois[:, :K].contiguous()))
    close(out["rcnn_cls"], ref["rcnn_cls"]); close(out["rcnn_reg"], ref["rcnn_reg"]); close(out["rcnn_feat"], ref["rcnn_feat"].squeeze(-1))
    eng.zero_grad(set_to_none=True)
    packed_rpn = {k: v for k, v in eng._folded.items() if not k.startswith(("xyz_up.", "merge_down", "rcnn_"))}
    assert packed_rpn
    loss = train_joint.rcnn_step(eng, xyz, img, xy, tids, opt, rois_per_frame=K, next_xyz=xyz)
    torch.cuda.synchronize()
    assert torch.isfinite(loss) and train_joint.LAST_GRAD_COLLECTIVES == 0
    named = dict(eng.named_parameters())
    for n in rcnn_names:
        assert named[n].grad is not None and bool(torch.isfinite(named[n].grad).all()), n
    stuck = [n for n in rcnn_names if torch.equal(before[n], named[n].detach())]
    # (the link head's last bias: the dual softmax is invariant to a constant added to every score — gradient zero up to rounding)
    assert len(stuck) <= 1, stuck
    for n, p in named.items():
        if not n.startswith("rcnn_net."):
            assert p.grad is None and torch.equal(before[n], p.detach()), n
    # the engine after the step: RCNN re-packed from the updated weights, the RPN's packed weights untouched (same objects)
    with torch.no_grad():
        rpn_out = eng.rpn_forward(xyz, img, xy)
        pts = eng.roi_pool(rpn_out, out["rois"])
        fused = eng.rcnn_forward(pts)
        want = train_joint.rcnn_forward_train(eng.rcnn_net, pts)
    close(fused["rcnn_cls"], want["rcnn_cls"]); close(fused["rcnn_reg"], want["rcnn_reg"])
    assert float((fused["rcnn_cls"] - ref["rcnn_cls"]).abs().max()) > 0          # (the step changed the network)
    for k, v in packed_rpn.items():
        assert eng._folded.get(k) is v, k


def test_rcnn_step_with_the_next_batchs_frozen_half_issued_ahead(tiny):
    """rcnn_step(next_batch=...): the next step's RPN forward / proposals / RoI pooling run on a stream of their own under this
    step's RCNN — same losses and the same weights after four steps over alternating batches as the plain step (the frozen half does
    not depend on the update; the library convolutions of the image branch are reproducible to ~1e-6 only, hence 1e-4 and not
    equality); a batch other than the announced one is computed in line"""
    from jmodt_amd import train_joint
    from jmodt_amd.detector import DetectorConfig
    from tests.test_gpu_detector import make_engine
    _, xyz, img, xy = tiny
    xyz2, img2, xy2 = synth.frames(2, 2048, 78, H=96, W=320, native=(94, 310))
    xy2 = np.random.default_rng(6).uniform(-0.98, 0.98, size=xy2.shape).astype(np.float32)
    batches = [(xyz, img, xy), (T(xyz2), T(img2), T(xy2))]
    K = 16
    tids = torch.randint(0, 6, (2, K), generator=torch.Generator().manual_seed(4)).float().to(DEV)
    runs = {}
    for mode in ("plain", "ahead", "wrong"):
        eng = make_engine(seed=3, cfg=DetectorConfig.tiny()).to(DEV)
        train_joint.prepare_rcnn(eng)
        for m in eng.modules():
            if isinstance(m, torch.nn.Dropout):
                m.eval()
        # (plain SGD: Adam's g / sqrt(v) turns a 1e-7 difference of a near-zero gradient into a full-size step)
        opt = torch.optim.SGD(eng.rcnn_net.parameters(), lr=1e-4)
        losses = []
        for i in range(4):
            cur, nxt = batches[i % 2], batches[(i + 1) % 2]
            if mode == "wrong":                  # announces the batch it is NOT given next
                nxt = cur
            losses.append(train_joint.rcnn_step(eng, *cur, tids, opt, rois_per_frame=K, next_batch=None if mode == "plain" else nxt))
            assert (getattr(eng, "_rcnn_ahead", None) is not None) == (mode != "plain")
        torch.cuda.synchronize()
        runs[mode] = ([float(x) for x in losses], {k: v.detach().clone() for k, v in eng.rcnn_net.named_parameters()})
    base_l, base_w = runs["plain"]
    assert len(set(round(x, 3) for x in base_l)) > 1                    # (the batches differ and the weights move)
    for mode in ("ahead", "wrong"):
        ls, ws = runs[mode]
        for a, b in zip(ls, base_l):
            assert abs(a - b) <= 1e-4 * max(1.0, abs(b)), (mode, ls, base_l)
        for k, w in base_w.items():
            close(ws[k], w, 1e-4)
