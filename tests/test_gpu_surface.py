"""GPU tier (-m gpu): operator-API surface that had no direct oracle test in round 1 — QueryAndGroup composition
(a5), the materialised-row affinity head `mlp3_forward` (a14), and the TRAINING affinity + finetune step on the
device (a16: rcnn.py:204-287, train_functions.py:282-329) against float64 restatements written from the
reference's own statements."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from jmodt_amd import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.mark.parametrize("use_xyz,C", [(True, 7), (True, 0), (False, 5)])
def test_query_and_group_vs_oracle(oracle, use_xyz, C):
    """QueryAndGroup.forward (pointnet2_utils.py:231-264) = ball_query + grouping_operation(xyz) - centre (+ cat with
    grouping_operation(features)): copies and one float32 subtraction, so bit-exact against the oracle composition"""
    from jmodt_amd.ops.pointnet2.pointnet2_utils import QueryAndGroup
    xyz = synth.dense_cloud(2, 500, 31, extent=5.0)
    new_xyz = np.ascontiguousarray(xyz[:, ::7][:, :64])
    feats = np.random.default_rng(3).normal(size=(2, C, 500)).astype(np.float32) if C else None
    got = QueryAndGroup(0.9, 16, use_xyz=use_xyz)(T(xyz), T(new_xyz), T(feats) if C else None).cpu().numpy()
    nb = oracle.ball_query(0.9, 16, xyz, new_xyz)
    gx = oracle.grouping_operation(np.ascontiguousarray(xyz.transpose(0, 2, 1)), nb) - new_xyz.transpose(0, 2, 1)[..., None]
    if C:
        gf = oracle.grouping_operation(feats, nb)
        want = np.concatenate([gx, gf], axis=1) if use_xyz else gf
    else:
        want = gx
    assert got.shape == want.shape and np.array_equal(got, want)


@pytest.mark.parametrize("M,C,H", [(300, 512, 512), (1, 64, 96), (4097, 128, 64), (5000, 64, (64, 32)), (9001, 32, (32, 7)),
                                   (130, 64, (64, 32))])
def test_mlp3_forward_values(M, C, H):
    """jm_mlp3_forward (the affinity head on materialised rows, rcnn.py:272-285) vs a float64 numpy MLP.  h2 <= 32 on the
    large-M path: the 128-column tiles write two projection-partial slots where divup(h2, 32) is one (the scratch is sized for
    both kernels; an undersized one let M floats land past the workspace)"""
    from jmodt_amd.ops.affinity import make_affinity_mlp, mlp3_forward
    torch.manual_seed(M)
    head = make_affinity_mlp(C, H if isinstance(H, tuple) else (H, H)).to(DEV).eval()
    with torch.no_grad():
        for m in head.modules():
            if isinstance(m, torch.nn.Conv1d):
                m.bias.normal_(0, 0.1)
    x = torch.relu(torch.randn(M, C, device=DEV))
    y = mlp3_forward(x, head).cpu().numpy()
    w = [t.detach().double().cpu().numpy() for t in (head[0].conv.weight[..., 0], head[0].conv.bias, head[2].conv.weight[..., 0],
                                                     head[2].conv.bias, head[3].conv.weight.reshape(-1), head[3].conv.bias)]
    h = np.maximum(x.double().cpu().numpy() @ w[0].T + w[1], 0)
    h = np.maximum(h @ w[2].T + w[3], 0)
    want = h @ w[4] + w[5]
    assert y.shape == (M,) and np.abs(y - want).max() < 1e-4
    # and the module itself (torch Conv1d path) agrees
    with torch.no_grad():
        assert (head(x.unsqueeze(-1)).flatten().cpu().numpy() - want).__abs__().max() < 1e-4


def _reference_training_affinity(feats, tids, link, se):
    """rcnn.py:204-287 restated statement by statement (float64, CPU), incl. get_unique_tid_feature (:145-156)"""
    def unique_tid_feature(fg_tid, fg_feat):
        diff = torch.min(fg_tid)
        clip = (fg_tid - diff).long()
        m = fg_tid.new_zeros(int(torch.max(clip)) + 1, len(fg_tid))
        m[clip, torch.arange(len(fg_tid))] = 1
        m = F.normalize(m, p=1, dim=1)
        mean = torch.mm(m, fg_feat)
        uniq = torch.unique(clip)
        return uniq + diff, mean[uniq]
    num_frames = tids.shape[0]
    prev_t, next_t = tids[range(0, num_frames, 2)], tids[range(1, num_frames, 2)]
    prev_f, next_f = feats[range(0, num_frames, 2)], feats[range(1, num_frames, 2)]
    out = dict(rcnn_link=[], gt_links=[], starts=[], ends=[], gt_starts=[], gt_ends=[])
    for i in range(num_frames // 2):
        pm, nm = prev_t[i] > 0, next_t[i] > 0
        if pm.sum() > 0 and nm.sum() > 0:
            ptid, pfeat = unique_tid_feature(prev_t[i][pm], prev_f[i][pm])
            ntid, nfeat = unique_tid_feature(next_t[i][nm], next_f[i][nm])
            ul = (ptid.unsqueeze(1) == ntid).double()
            cor = torch.abs(pfeat.unsqueeze(1).repeat(1, len(ntid), 1) - nfeat.unsqueeze(0).repeat(len(ptid), 1, 1))
            s = link(cor.view(len(ptid) * len(ntid), -1, 1)).view(len(ptid), len(ntid))
            s = (torch.softmax(s, dim=1) + torch.softmax(s, dim=0)) / 2
            out["rcnn_link"].append(s.view(-1, 1)); out["gt_links"].append(ul.view(-1))
            out["gt_starts"].append(1 - ul.sum(0)); out["gt_ends"].append(1 - ul.sum(1))
            out["starts"].append(cor.mean(dim=0)); out["ends"].append(cor.mean(dim=1))
    return dict(rcnn_link=torch.cat(out["rcnn_link"]), gt_links=torch.cat(out["gt_links"]),
                rcnn_start=se(torch.cat(out["starts"]).unsqueeze(-1)).squeeze(-1), gt_starts=torch.cat(out["gt_starts"]),
                rcnn_end=se(torch.cat(out["ends"]).unsqueeze(-1)).squeeze(-1), gt_ends=torch.cat(out["gt_ends"]))


def _reid_loss_reference(o):
    """train_functions.py:282-329 with LOSS_LINK = LOSS_SE = 'L1' and unit weights"""
    return (F.l1_loss(o["rcnn_link"].view(-1), o["gt_links"]) + F.l1_loss(torch.sigmoid(o["rcnn_start"].view(-1)), o["gt_starts"])
            + F.l1_loss(torch.sigmoid(o["rcnn_end"].view(-1)), o["gt_ends"]))


def test_training_affinity_and_finetune_step_on_gpu():
    import copy
    from jmodt_amd.ops.affinity import make_affinity_mlp
    from jmodt_amd.ops.affinity_train import finetune_step, reid_loss, training_affinity
    g = torch.Generator().manual_seed(5)
    frames, R, C = 6, 64, 512
    feats = torch.relu(torch.randn(frames, R, C, generator=g))
    tids = torch.randint(0, 9, (frames, R), generator=g).float()
    tids[2] = 0                                   # a pair without foreground on one side is skipped (rcnn.py:230)
    torch.manual_seed(1)
    link, se = make_affinity_mlp(), make_affinity_mlp()
    with torch.no_grad():
        for m in list(link.modules()) + list(se.modules()):
            if isinstance(m, torch.nn.Conv1d):
                m.bias.normal_(0, 0.05)
    link64, se64 = copy.deepcopy(link).double(), copy.deepcopy(se).double()
    want = _reference_training_affinity(feats.double(), tids.double(), link64, se64)
    dlink, dse = copy.deepcopy(link).to(DEV).train(), copy.deepcopy(se).to(DEV).train()
    got = training_affinity(feats.to(DEV), tids.to(DEV), dlink, dse)
    for k in want:
        assert got[k].shape == want[k].shape, k
        assert (got[k].double().cpu() - want[k]).abs().max().item() < 1e-4, k
    assert abs(reid_loss(got).item() - _reid_loss_reference(want).item()) < 1e-5
    # one finetune step (Adam, single rank): loss and updated parameters vs the float64 CPU step
    opt = torch.optim.Adam(list(dlink.parameters()) + list(dse.parameters()), lr=1e-3)
    loss = finetune_step(feats.to(DEV), tids.to(DEV), dlink, dse, opt, world=1)
    opt64 = torch.optim.Adam(list(link64.parameters()) + list(se64.parameters()), lr=1e-3)
    opt64.zero_grad()
    l64 = _reid_loss_reference(_reference_training_affinity(feats.double(), tids.double(), link64, se64))
    l64.backward()
    opt64.step()
    assert abs(loss - l64.item()) < 1e-5
    moved = 0.0
    for pd, p64, p0 in zip(list(dlink.parameters()) + list(dse.parameters()), list(link64.parameters()) + list(se64.parameters()),
                           list(link.parameters()) + list(se.parameters())):
        # Adam's first step is lr * sign(grad) wherever |grad| >> eps: compare where the float64 gradient is not tiny
        big = p64.grad.abs() > 1e-7
        if big.any():
            assert (pd.detach().double().cpu() - p64.detach())[big].abs().max().item() < 2e-4
        moved = max(moved, (pd.detach().cpu() - p0).abs().max().item())
    assert moved > 5e-4
    # the static-shape, sync-free step (what bench.py --workload train runs): same loss, same update
    from jmodt_amd.ops.affinity_train import finetune_step_static
    slink, sse = copy.deepcopy(link).to(DEV).train(), copy.deepcopy(se).to(DEV).train()
    sopt = torch.optim.Adam(list(slink.parameters()) + list(sse.parameters()), lr=1e-3)
    sloss = finetune_step_static(feats.to(DEV), tids.to(DEV), slink, sse, sopt, world=1)
    assert sloss.is_cuda and abs(sloss.item() - l64.item()) < 1e-5
    for ps, p64 in zip(list(slink.parameters()) + list(sse.parameters()), list(link64.parameters()) + list(se64.parameters())):
        big = p64.grad.abs() > 1e-7
        if big.any():
            assert (ps.detach().double().cpu() - p64.detach())[big].abs().max().item() < 2e-4


@pytest.mark.parametrize("frames,R,C,H,ntid", [(6, 64, 512, 512, 9), (4, 64, 512, 512, 30), (2, 48, 64, 64, 5), (8, 128, 128, 96, 40),
                                               (8, 128, 64, 32, 40)])     # h2 = 32 on 65536 pair rows: two partial slots per tile
def test_training_affinity_hip_kernels_vs_float64_reference(frames, R, C, H, ntid):
    """a16 on the matrix cores (csrc/affinity_train.hip): outputs, loss and the gradients of all twelve head tensors
    against (i) the float64 statement-by-statement copy of rcnn.py:204-287 + train_functions.py:282-329 differentiated by
    torch autograd on the CPU, (ii) the plain-torch static form.  Bars: 1e-4 on link / sigmoid outputs and the loss,
    1e-4 x max|grad| per gradient tensor."""
    import copy
    from jmodt_amd.ops.affinity import make_affinity_mlp
    from jmodt_amd.ops.affinity_train import (AffinityTrainState, affinity_train_loss, reid_loss_static, training_affinity_hip,
                                              training_affinity_static)
    g = torch.Generator().manual_seed(frames * R + C)
    feats = torch.relu(torch.randn(frames, R, C, generator=g))
    tids = torch.randint(0, ntid, (frames, R), generator=g).float()
    if frames >= 6:
        tids[2] = 0                                # a pair without foreground on one side is skipped (rcnn.py:230)
    torch.manual_seed(3)
    link, se = make_affinity_mlp(C, (H, H)), make_affinity_mlp(C, (H, H))
    with torch.no_grad():
        for m in list(link.modules()) + list(se.modules()):
            if isinstance(m, torch.nn.Conv1d):
                m.bias.normal_(0, 0.05)
    link64, se64 = copy.deepcopy(link).double(), copy.deepcopy(se).double()
    f64 = feats.double().requires_grad_(True)
    want = _reference_training_affinity(f64, tids.double(), link64, se64)
    l64 = _reid_loss_reference(want)
    l64.backward()
    dlink, dse = copy.deepcopy(link).to(DEV).train(), copy.deepcopy(se).to(DEV).train()
    f_d, t_d = feats.to(DEV), tids.to(DEV)
    out = training_affinity_hip(f_d, t_d, dlink, dse)
    # (i) outputs: the reference's per-pair tensors are the slot matrices restricted to the representatives; its rows are
    # ordered by ascending track id (torch.unique), ours by slot — compare as sorted multisets per pair via the static torch form
    ref = training_affinity_static(f_d, t_d, dlink, dse)
    for k in ("valid", "start_valid", "end_valid"):
        assert torch.equal(out[k], ref[k]), k
    v, sv, ev = ref["valid"], ref["start_valid"], ref["end_valid"]
    assert int(v.sum()) == want["gt_links"].numel() and int(sv.sum()) == want["gt_starts"].numel()
    assert torch.equal(out["gt_links"], ref["gt_links"]) and torch.equal(out["gt_starts"] * sv, ref["gt_starts"] * sv)
    assert torch.equal(out["gt_ends"] * ev, ref["gt_ends"] * ev)
    assert (out["link"] - ref["link"] * v).abs().max().item() < 1e-5
    assert ((torch.sigmoid(out["start"]) - torch.sigmoid(ref["start"])) * sv).abs().max().item() < 1e-5
    assert ((torch.sigmoid(out["end"]) - torch.sigmoid(ref["end"])) * ev).abs().max().item() < 1e-5
    assert abs(out["loss"].item() - l64.item()) < 1e-5
    assert abs(reid_loss_static(ref)[0].item() - l64.item()) < 1e-5
    got_sorted = torch.sort(out["link"][v].double().cpu())[0]
    assert (got_sorted - torch.sort(want["rcnn_link"].view(-1))[0]).abs().max().item() < 1e-4
    # (ii) gradients through the autograd.Function
    f_g = f_d.clone().requires_grad_(True)          # (joint training: the features' gradient comes back as well)
    st = AffinityTrainState(f_g, t_d)
    loss = affinity_train_loss(st, dlink, dse)
    assert abs(loss.item() - l64.item()) < 1e-5
    (2.0 * loss).backward()
    gf = 2.0 * f64.grad
    assert (f_g.grad.double().cpu() - gf).abs().max().item() <= 1e-4 * gf.abs().max().item() + 1e-9 and gf.abs().max().item() > 0
    names = ["w1", "b1", "w2", "b2", "w3", "b3"]
    for head, (pd_list, p64_list) in (("link", (list(dlink.parameters()), list(link64.parameters()))),
                                      ("se", (list(dse.parameters()), list(se64.parameters())))):
        for nm, pd, p64 in zip(names, pd_list, p64_list):
            gw = 2.0 * p64.grad
            scale = max(gw.abs().max().item(), 1e-12)
            err = (pd.grad.double().cpu() - gw).abs().max().item()
            # (the link head's b3 gradient is ZERO mathematically: both softmaxes are invariant to a constant shift of the scores)
            assert err <= 1e-4 * scale + 1e-7, (head, nm, err, scale)
            assert gw.abs().max().item() > 0 or (head, nm) == ("link", "b3")
    # determinism: the split-M partial sums and the projection's column-group partials are reduced in a fixed order (no float
    # atomics anywhere on the path): three repeats are BIT-identical in all twelve gradient tensors and the loss
    runs = []
    for _ in range(3):
        dlink.zero_grad(); dse.zero_grad()
        ls = affinity_train_loss(AffinityTrainState(f_d, t_d), dlink, dse)
        (2.0 * ls).backward()
        runs.append([ls.detach().clone()] + [p.grad.clone() for p in list(dlink.parameters()) + list(dse.parameters())])
    for other in runs[1:]:
        assert all(torch.equal(a, b) for a, b in zip(runs[0], other))


def test_finetune_step_static_uses_the_hip_kernels():
    """bench.py --workload train's step: on GPU tensors it is the hand-written kernels that run (profiler records)"""
    from jmodt_amd.ops.affinity import make_affinity_mlp
    from jmodt_amd.ops.affinity_train import finetune_step_static
    from jmodt_amd.profile import prof
    torch.manual_seed(0)
    link, se = make_affinity_mlp().to(DEV).train(), make_affinity_mlp().to(DEV).train()
    opt = torch.optim.Adam(list(link.parameters()) + list(se.parameters()), lr=1e-3)
    g = torch.Generator().manual_seed(1)
    feats = torch.relu(torch.randn(4, 64, 512, generator=g)).to(DEV)
    tids = torch.randint(0, 13, (4, 64), generator=g).float().to(DEV)
    before = [p.detach().clone() for p in link.parameters()]
    prof.reset(); prof.enabled = True
    try:
        loss = finetune_step_static(feats, tids, link, se, opt, world=1)
        torch.cuda.synchronize()
        names = set(prof.records)
    finally:
        prof.enabled = False; prof.reset()
    assert {"affinity_train_prepare", "affinity_train_link_step", "affinity_train_se_step"} <= names, names
    assert loss.is_cuda and 0 < loss.item() < 3
    assert all((a - b.detach()).abs().max().item() > 0 for a, b in zip(before, link.parameters()))


def test_gradient_allreduce_runs_on_rccl_with_one_rank(tmp_path):
    """flatten -> RCCL all_reduce -> unflatten of the 4 206 600-byte finetune gradient on DEVICE tensors, with a one-rank
    process group (tools/train.py:86-88's DataParallel reduction as one process per GPU): the collective is issued and timed,
    and three Adam steps through it leave exactly the parameters of three steps without any process group.  The first N > 1
    run on an 8-GPU node then executes no code this box has not."""
    import subprocess
    import sys
    helper = os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers", "rccl_one_rank_step.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    res = {}
    for tag, extra in (("plain", []), ("group", ["--group"])):
        out = str(tmp_path / f"{tag}.pt")
        p = subprocess.run([sys.executable, helper, out] + extra, capture_output=True, text=True, timeout=600, env=env)
        assert p.returncode == 0, p.stderr[-3000:]
        res[tag] = torch.load(out)
    assert res["plain"]["issued"] == 0 and res["group"]["issued"] == 1
    assert res["group"]["bytes_per_step"] == 4_206_600 and res["group"]["ms_per_step"] > 0
    assert res["group"]["losses"] == res["plain"]["losses"]
    for a, b in zip(res["group"]["params"], res["plain"]["params"]):
        assert torch.equal(a, b)


def test_finetune_step_hip_data_parallel_equals_single_process(tmp_path):
    """the finetune step on the training KERNELS under world size 2: each rank back-propagates its local sums over the GLOBAL
    element counts (all-reduced on the device), the flat gradient bucket is summed, and the parameters after the step equal one
    process on the whole batch — the CPU tier shows this for the plain-torch form (tests/test_dist_cpu.py); here the hand-written
    kernels and device-tensor collectives run it (two ranks on cuda:0 over gloo: RCCL cannot share a device)"""
    import socket
    import subprocess
    import sys
    helper = os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers", "finetune_dp_worker.py")
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = [subprocess.Popen([sys.executable, helper, str(tmp_path / f"r{r}.pt")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              env=dict(base, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)))
             for r in range(2)]
    one = subprocess.run([sys.executable, helper, str(tmp_path / "one.pt"), "--single"], capture_output=True, text=True, timeout=600, env=base)
    assert one.returncode == 0, one.stderr[-3000:]
    for p in procs:
        _, err = p.communicate(timeout=600)
        assert p.returncode == 0, err[-3000:]
    ref, r0, r1 = (torch.load(tmp_path / f) for f in ("one.pt", "r0.pt", "r1.pt"))
    assert r0["frames"] == (0, 4) and r1["frames"] == (4, 8)
    assert abs(r0["loss"] - ref["loss"]) < 1e-5 and abs(r1["loss"] - ref["loss"]) < 1e-5      # the all-reduced whole-batch loss
    for a, b, want in zip(r0["params"], r1["params"], ref["params"]):
        assert torch.equal(a, b)                                                          # replicas stay identical
        assert float((a - want).abs().max()) <= 1e-6 + 1e-5 * float(want.abs().max())       # DP == single process


# ------------------------------------------------------------------ contraction-proof decision fixtures
# Index outputs depend on comparisons of float32 expressions whose rounding depends on whether the compiler contracts
# a*b + c into an FMA (DESIGN.md §3 states the two conventions used).  These inputs make every compared quantity EXACT
# in float32 under ANY contraction (small dyadic rationals), and put points / boxes exactly ON the decision boundary.
def test_ball_query_points_exactly_at_the_radius_grid(oracle):
    """coordinates on a 2^-3 grid, radius 0.5 (r^2 = 0.25 exact): points at distance exactly r are NOT neighbours
    (strict <, ball_query_gpu.cu:33-35), points one grid step inside are"""
    from jmodt_amd.ops.pointnet2.pointnet2_utils import ball_query
    g = np.arange(-8, 9, dtype=np.float32) / 8.0
    pts = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(1, -1, 3).astype(np.float32)       # 17^3 lattice
    centres = np.array([[[0, 0, 0], [0.5, 0.5, 0.5], [-1, -1, -1], [0.125, -0.25, 0.375]]], np.float32)
    for ns in (16, 64):
        got = ball_query(0.5, ns, T(pts), T(centres)).cpu().numpy()
        assert np.array_equal(got, oracle.ball_query(0.5, ns, pts, centres))
    d2 = ((pts[0][None] - centres[0][:, None]) ** 2).sum(-1)                                           # exact
    full = ball_query(0.5, 64, T(pts), T(centres)).cpu().numpy()[0]
    for c in range(4):
        inside = np.nonzero(d2[c] < 0.25)[0]
        on_sphere = np.nonzero(d2[c] == 0.25)[0]
        assert len(on_sphere) >= 6 or c == 2
        hits = np.unique(full[c])
        assert set(hits) <= set(inside) and not (set(hits) & set(on_sphere))
        assert np.array_equal(np.sort(full[c][:min(64, len(inside))]), inside[:64])


def test_roipool_points_exactly_on_box_faces_grid(oracle):
    """axis-aligned boxes (ry = 0: cos = 1, sin = 0 exactly) with dyadic sizes on a lattice cloud: points exactly on a
    face belong to the box (closed intervals, roipool3d_kernel.cu:14-28); the pooled index order is the point order"""
    from jmodt_amd.ops.roipool3d.roipool3d_utils import roipool3d_gpu
    g = np.arange(-8, 9, dtype=np.float32) / 4.0
    pts = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(1, -1, 3).astype(np.float32)
    feat = np.arange(pts.shape[1], dtype=np.float32).reshape(1, -1, 1)                                 # feature = point index
    # [x, y(bottom), z, h, w, l, ry]; after the 0.25 enlargement every face lies ON lattice planes
    boxes = np.array([[[0, 0.5, 0, 0.5, 0.5, 1.0, 0], [1, 1, -1, 1.5, 1.5, 0.5, 0], [-1.5, 0, 1.5, 0.0, 0.5, 0.5, 0]]], np.float32)
    pooled, empty = roipool3d_gpu(T(pts), T(feat), T(boxes), 0.25, 64)
    want, wempty = oracle.roipool3d(pts, feat, oracle.enlarge_box3d(boxes, 0.25), 64)
    assert np.array_equal(pooled.cpu().numpy(), want) and np.array_equal(empty.cpu().numpy(), wempty)
    e = oracle.enlarge_box3d(boxes, 0.25)[0]
    for m in range(3):
        cx, by, cz, h, w, l = e[m, :6]
        inside = (np.abs(pts[0, :, 0] - cx) <= l / 2) & (np.abs(pts[0, :, 1] - (by - h / 2)) <= h / 2) & (np.abs(pts[0, :, 2] - cz) <= w / 2)
        idx = np.nonzero(inside)[0]
        on_face = (np.abs(pts[0, idx, 0] - cx) == l / 2) | (np.abs(pts[0, idx, 2] - cz) == w / 2)
        assert on_face.any()
        assert np.array_equal(pooled.cpu().numpy()[0, m, :min(64, len(idx)), 3], idx[:64].astype(np.float32))


def test_nms_normal_iou_exactly_at_the_threshold(oracle):
    """integer boxes whose axis-aligned IoU is exactly 1/2, 1/4, 3/4: suppression is strict (iou > thresh,
    iou3d_kernel.cu:318-330), so a pair AT the threshold is kept, one ulp below the threshold value it is suppressed"""
    from jmodt_amd.ops.iou3d.iou3d_utils import nms_normal_gpu
    boxes = np.array([[0, 0, 2, 1, 0], [0, 0, 1, 1, 0],          # IoU 1/2 with box 0
                      [10, 0, 14, 1, 0], [10, 0, 11, 1, 0],      # IoU 1/4
                      [20, 0, 24, 1, 0], [20, 0, 23, 1, 0],      # IoU 3/4
                      [30, 0, 31, 1, 0], [30, 0, 31, 1, 0]], np.float32)    # IoU 1
    scores = np.array([0.9, 0.8, 0.7, 0.6, 0.5, 0.4, 0.3, 0.2], np.float32)
    for thr, kept in ((0.5, [0, 1, 2, 3, 4, 6]), (0.25, [0, 2, 3, 4, 6]), (0.75, [0, 1, 2, 3, 4, 5, 6]),
                      (float(np.nextafter(np.float32(0.5), np.float32(0))), [0, 2, 3, 4, 6])):
        got = nms_normal_gpu(T(boxes), T(scores), thr).cpu().numpy()
        assert np.array_equal(got, oracle.nms(boxes, scores, thr, normal=True))
        assert got.tolist() == kept, (thr, got)


@pytest.mark.parametrize("nb,P,D,C", [(8, 128, 128, 512), (3, 7, 5, 64), (2, 1, 33, 96), (1, 40, 40, 512)])
def test_pairwise_affinity_batched_matches_per_problem(oracle, nb, P, D, C):
    """jm_affinity_forward_batched / _start_end_batched (every frame pair of a batch as one GEMM chain) vs the
    single-problem entry and vs the oracle"""
    from jmodt_amd.ops.affinity import make_affinity_mlp, pairwise_affinity, pairwise_affinity_batched
    torch.manual_seed(nb * P + D)
    link, se = make_affinity_mlp(C, (C, C)).to(DEV).eval(), make_affinity_mlp(C, (C, C)).to(DEV).eval()
    with torch.no_grad():
        for m in list(link.modules()) + list(se.modules()):
            if isinstance(m, torch.nn.Conv1d):
                m.bias.normal_(0, 0.05)
    pf = torch.relu(torch.randn(nb, P, C, device=DEV))
    df = torch.relu(torch.randn(nb, D, C, device=DEV))
    A, s, e, raw = pairwise_affinity_batched(pf, df, link, se, return_raw=True)
    assert A.shape == (nb, P, D) and s.shape == (nb, D) and e.shape == (nb, P)

    def w(h):
        return tuple(a.detach().cpu().numpy().copy() for a in (
            h[0].conv.weight[..., 0], h[0].conv.bias, h[2].conv.weight[..., 0], h[2].conv.bias,
            h[3].conv.weight.reshape(-1), h[3].conv.bias))
    for b in range(nb):
        A1, s1, e1, raw1 = pairwise_affinity(pf[b], df[b], link, se, return_raw=True)
        for got, one in ((A[b], A1), (s[b], s1), (e[b], e1), (raw[b], raw1)):
            assert (got - one).abs().max().item() < 2e-5
        if b < 2:
            oA, os_, oe = oracle.affinity(pf[b].cpu().numpy(), df[b].cpu().numpy(), w(link), w(se))
            assert np.abs(A[b].cpu().numpy() - oA).max() < 1e-4 and np.abs(s[b].cpu().numpy() - os_).max() < 1e-4
            assert np.abs(e[b].cpu().numpy() - oe).max() < 1e-4
    # bit-reproducible: the projection partials are summed in slot order (no float atomics)
    for _ in range(3):
        A2, s2, e2, raw2 = pairwise_affinity_batched(pf, df, link, se, return_raw=True)
        assert torch.equal(raw2, raw) and torch.equal(A2, A) and torch.equal(s2, s) and torch.equal(e2, e)
    A3, s3, e3, raw3 = pairwise_affinity(pf[0], df[0], link, se, return_raw=True)
    A4, s4, e4, raw4 = pairwise_affinity(pf[0], df[0], link, se, return_raw=True)
    assert torch.equal(raw3, raw4) and torch.equal(A3, A4) and torch.equal(s3, s4) and torch.equal(e3, e4)


@pytest.mark.parametrize("nb,P,D,C", [(2, 37, 50, 128), (1, 65, 63, 512), (3, 128, 128, 64), (5, 3, 200, 256), (1, 1, 1, 512)])
def test_link_head_one_kernel_form_vs_float64(nb, P, D, C):
    """csrc/affinity_fused.hip (both hidden layers of the 512-512 link head in one kernel, hidden activation in LDS) on shapes
    whose 64-row tiles end inside a problem, cross prediction rows and problem boundaries, and on every supported input width:
    raw scores against a float64 evaluation of the reference's module (rcnn.py:239-258), the dual softmax on top, run-to-run bits"""
    from jmodt_amd.ops.affinity import make_affinity_mlp, pairwise_affinity_batched
    torch.manual_seed(nb * 1000 + P * 10 + D)
    link = make_affinity_mlp(C, (512, 512)).to(DEV).eval()
    with torch.no_grad():
        for m in link.modules():
            if isinstance(m, torch.nn.Conv1d):
                m.bias.normal_(0, 0.05)
    pf = torch.relu(torch.randn(nb, P, C, device=DEV))
    df = torch.relu(torch.randn(nb, D, C, device=DEV))
    A, raw = pairwise_affinity_batched(pf, df, link, None, return_raw=True)
    cor = (pf[:, :, None, :] - df[:, None, :, :]).abs().double().reshape(nb, P * D, C).transpose(1, 2)
    want = link.double()(cor).reshape(nb, P, D)
    link.float()
    scale = max(1.0, want.abs().max().item())
    assert (raw.double() - want).abs().max().item() < 1e-4 * scale              # (measured: 3e-7)
    sm = (torch.softmax(want, 2) + torch.softmax(want, 1)) / 2
    assert (A.double() - sm).abs().max().item() < 1e-4
    for _ in range(2):
        A2, raw2 = pairwise_affinity_batched(pf, df, link, None, return_raw=True)
        assert torch.equal(raw2, raw) and torch.equal(A2, A)


# ------------------------------------------------------------------ EXPERIMENTAL split-bf16 affinity (csrc/affinity_x3.hip)
@pytest.mark.parametrize("nb,P,D,C", [(8, 128, 128, 512), (2, 64, 64, 512), (1, 37, 50, 64)])
def test_affinity_split_bf16_error_not_above_exact_fp32(nb, P, D, C):
    """opt-in path only: every fp32 product as six bf16 products (3-term splits).  Its error against a float64 evaluation must
    be AT THE LEVEL of the exact-fp32 MFMA kernel's (both ~1e-6 on O(1) scores, i.e. fp32 rounding of a 512-term sum: the two
    maxima differ by a few per cent either way, which is noise of the maximum, so the bar is 1.25 x), and the final affinity stays
    within the 1e-4 bar with a 10 x margin."""
    from jmodt_amd.ops.affinity import make_affinity_mlp, pairwise_affinity_batched
    torch.manual_seed(nb * P + C)
    link = make_affinity_mlp(C, (C, C)).to(DEV).eval()
    with torch.no_grad():
        for m in link.modules():
            if isinstance(m, torch.nn.Conv1d):
                m.bias.normal_(0, 0.05)
    g = torch.Generator().manual_seed(3)
    pf = torch.relu(torch.randn(nb, P, C, generator=g)).to(DEV)
    df = torch.relu(torch.randn(nb, D, C, generator=g)).to(DEV)
    A32, raw32 = pairwise_affinity_batched(pf, df, link, None, return_raw=True)
    Ax3, rawx3 = pairwise_affinity_batched(pf, df, link, None, return_raw=True, split_bf16=True)
    w = [t.detach().double() for t in (link[0].conv.weight[..., 0], link[0].conv.bias, link[2].conv.weight[..., 0], link[2].conv.bias,
                                       link[3].conv.weight.reshape(-1), link[3].conv.bias)]
    cor = (pf.double().unsqueeze(2) - df.double().unsqueeze(1)).abs()                       # (nb, P, D, C)
    h = torch.relu(cor @ w[0].t() + w[1])
    h = torch.relu(h @ w[2].t() + w[3])
    want = h @ w[4] + w[5]
    e32 = (raw32.double() - want).abs().max().item()
    ex3 = (rawx3.double() - want).abs().max().item()
    print(f"raw link scores, max |err| vs float64: exact fp32 MFMA {e32:.3e}, split bf16 (6 products) {ex3:.3e}, |S|max {want.abs().max().item():.2f}")
    assert ex3 <= max(e32 * 1.25, 1e-6), (ex3, e32)
    wantA = (torch.softmax(want, 2) + torch.softmax(want, 1)) / 2
    assert (Ax3.double() - wantA).abs().max().item() < 1e-5 and (A32.double() - wantA).abs().max().item() < 1e-5


def test_training_affinity_hip_kernels_vs_the_references_train_forward_and_loss():
    """a16 on the matrix cores against train_ref.npz — the reference's own PointRCNN.forward in TRAIN mode, its re-id loss
    (get_rcnn_loss under FINETUNE) and the gradients its autograd produced (tests/golden/make_golden_train.py): per-pair link
    scores as multisets (the reference orders a pair's rows by track id, the static form by RoI slot), start / end
    probabilities, the loss and all twelve gradient tensors"""
    from jmodt_amd.ops.affinity_train import AffinityTrainState, affinity_train_loss, training_affinity_hip
    from tests.test_oracle_cpu import reference_train_fixture
    g, link, se = reference_train_fixture()
    link, se = link.to(DEV).train(), se.to(DEV).train()
    feats, tids = torch.from_numpy(g["roi_feat"]).to(DEV), torch.from_numpy(g["gt_tids"]).to(DEV)
    w_link, w_se = float(g["weights"][0]), float(g["weights"][1])
    out = training_affinity_hip(feats, tids, link, se)
    v, sv, ev = out["valid"], out["start_valid"], out["end_valid"]
    assert int(v.sum()) == g["rcnn_link"].size and int(sv.sum()) == g["gt_starts"].size and int(ev.sum()) == g["gt_ends"].size
    assert (torch.sort(out["link"][v].cpu())[0] - torch.sort(torch.from_numpy(g["rcnn_link"]).view(-1))[0]).abs().max().item() < 1e-5
    assert (torch.sort(torch.sigmoid(out["start"][sv]).cpu())[0]
            - torch.sort(torch.sigmoid(torch.from_numpy(g["rcnn_start"]).view(-1)))[0]).abs().max().item() < 1e-5
    assert (torch.sort(torch.sigmoid(out["end"][ev]).cpu())[0]
            - torch.sort(torch.sigmoid(torch.from_numpy(g["rcnn_end"]).view(-1)))[0]).abs().max().item() < 1e-5
    assert int(out["gt_links"][v].sum()) == int(g["gt_links"].sum())
    feats.requires_grad_(True)                     # joint training: d(loss)/d(RoI features) comes back too
    loss = affinity_train_loss(AffinityTrainState(feats, tids), link, se, link_weight=w_link, se_weight=w_se)
    assert abs(loss.item() - float(g["loss"])) < 1e-5
    loss.backward()
    want = g["grad.roi_feat"]
    assert np.abs(feats.grad.cpu().numpy() - want).max() <= 1e-4 * np.abs(want).max() + 1e-9 and np.abs(want).max() > 1e-3
    assert (feats.grad[torch.from_numpy(g["gt_tids"]).to(DEV) <= 0] == 0).all()          # background RoIs get none
    for head, mod in (("link_layer", link), ("se_layer", se)):
        for k, p_ in mod.named_parameters():
            want = g[f"grad.rcnn_net.{head}.{k}"]
            got = p_.grad.cpu().numpy()
            assert np.abs(got - want).max() <= 1e-4 * max(np.abs(want).max(), 1e-12) + 1e-7, (head, k)


def test_inference_affinity_and_cost_vs_the_references_tracker_update():
    """a15 + f1 on the GPU against tracker_ref.npz: the arguments the reference's own Tracker.update (tracker.py:50-112) built
    for its solver — dual-softmax link scores, w_se * sigmoid(start / end), class scores — and the solver's cost matrix"""
    from jmodt_amd.ops.affinity import pairwise_affinity
    from jmodt_amd.ops.association import association_cost
    from tests.test_oracle_cpu import reference_train_fixture
    g, link, se = reference_train_fixture("tracker_ref.npz", "pred_feat")
    link, se = link.to(DEV).eval(), se.to(DEV).eval()
    w_cls, w_app, w_iou, w_dis, w_se = (float(v) for v in g["weights"])
    P, D = g["pred_feat"].shape[0], g["det_feat"].shape[0]
    G = lambda k: torch.from_numpy(g[k]).to(DEV)       # noqa: E731
    A, start, end = pairwise_affinity(G("pred_feat"), G("det_feat"), link, se)
    assert (A.cpu() - torch.from_numpy(g["solver.link"])).abs().max().item() < 1e-5
    new = torch.cat([torch.zeros(P), (w_se * torch.sigmoid(start)).cpu()])
    endv = torch.cat([(w_se * torch.sigmoid(end)).cpu(), torch.zeros(D)])
    assert (new.double() - torch.from_numpy(g["solver.new"])).abs().max().item() < 1e-5
    assert (endv.double() - torch.from_numpy(g["solver.end"])).abs().max().item() < 1e-5
    cost, iou, dist = association_cost(G("pred_boxes"), G("det_boxes"), A, w_app, w_iou, w_dis, return_parts=True)
    for got, key in ((cost, "solver.cost"), (iou, "solver.iou"), (dist, "solver.dis")):
        assert (got.cpu() - torch.from_numpy(g[key])).abs().max().item() < 2e-5, key
