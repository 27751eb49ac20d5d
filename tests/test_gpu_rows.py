"""GPU tier (-m gpu): the training path's row kernels (csrc/rows_gemm.hip, csrc/rows_ops.hip, ops/rows.py, train_rows.py) against
torch autograd in float64 on the same operands, and the whole rows route of the joint-mode step against the operator route
(train_joint.joint_forward: the composition already pinned to the reference's own autograd by backward_ref.npz)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from jmodt_amd import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def close(got, want, tol=1e-4, what=""):
    got, want = got.detach().double().cpu(), want.detach().double().cpu()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    scale = max(1.0, float(want.abs().max()))
    err = float((got - want).abs().max())
    assert err <= tol * scale, (what, err, scale)


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


@pytest.mark.parametrize("M,k1,k2,n,act", [(300, 64, 0, 76, 1), (1000, 196, 0, 256, 1), (257, 96, 128, 128, 1), (4096, 24, 24, 4, 2),
                                           (130, 8, 0, 128, 0), (5000, 128, 0, 196, 1), (64, 512, 512, 512, 1), (20000, 128, 0, 256, 1), (33000, 64, 36, 132, 1)])
def test_rows_linear_forward_dgrad_wgrad_vs_float64(M, k1, k2, n, act):
    """every GEMM mode on shapes with row / column / contraction tails (196, 24, 76, 4 = the widths of config.py:75-77 that are
    no multiple of 16), one and two operands, against float64; the last two shapes take the 128 x 128 persistent tiles, the others
    the one-wave 32 x 32 tiles (a device-side row count always takes the former)"""
    from jmodt_amd.ops import rows as R
    x1, x2 = rnd(M, k1, seed=1), (rnd(M, k2, seed=2) if k2 else None)
    w, b = rnd(n, k1 + k2, seed=3, scale=0.2), rnd(n, seed=4)
    y = R.linear_forward(x1, w, b, act, x2)
    xx = torch.cat([x1, x2], 1) if k2 else x1
    pre = xx.double() @ w.double().t() + b.double()
    want = {0: pre, 1: torch.relu(pre), 2: torch.tanh(pre)}[act]
    close(y, want, what="forward")
    dy = rnd(M, n, seed=5)
    mask = rnd(M, k1, seed=6)
    dx = R.linear_dgrad(dy, w, 0, k1, mask=mask)
    close(dx, (dy.double() @ w.double()[:, :k1]) * (mask > 0), what="dgrad masked")
    if k2:
        dx2 = R.linear_dgrad(dy, w, k1, k2)
        close(dx2, dy.double() @ w.double()[:, k1:], what="dgrad second operand")
        acc = rnd(M, k2, seed=7)
        want_acc = acc.double() + dy.double() @ w.double()[:, k1:]
        close(R.linear_dgrad(dy, w, k1, k2, accumulate_into=acc), want_acc, what="dgrad accumulate")
    dw, db = R.linear_wgrad(dy, [x1, x2] if k2 else [x1])
    close(dw, dy.double().t() @ xx.double(), tol=2e-4, what="wgrad")
    close(db, dy.double().sum(0), tol=2e-4, what="bias grad")
    # the row count in device memory: rows beyond it are neither read nor written
    mv = M // 2 + 3
    m_dev = torch.tensor([mv], dtype=torch.int32, device=DEV)
    y2 = R.linear_forward(x1, w, b, act, x2, m_dev=m_dev)
    close(y2[:mv], want[:mv], what="forward, device row count")
    dx3 = R.linear_dgrad(dy, w, 0, k1, mask=mask, m_dev=m_dev)
    close(dx3[:mv], ((dy.double() @ w.double()[:, :k1]) * (mask > 0))[:mv], what="dgrad, device row count")
    dw2, db2 = R.linear_wgrad(dy, [x1, x2] if k2 else [x1], m_dev=m_dev)
    close(dw2, dy[:mv].double().t() @ xx[:mv].double(), tol=2e-4, what="wgrad, device row count")
    close(db2, dy[:mv].double().sum(0), tol=2e-4, what="bias grad, device row count")
    assert torch.equal(dw2, R.linear_wgrad(dy, [x1, x2] if k2 else [x1], m_dev=m_dev)[0])           # fixed-order reduction


def test_rows_mlp_autograd_vs_float64():
    from jmodt_amd.ops import rows as R
    M = 777
    x1, x2 = rnd(M, 64, seed=1).requires_grad_(), rnd(M, 32, seed=2).requires_grad_()
    Ws = [rnd(128, 96, seed=3, scale=0.2).requires_grad_(), rnd(196, 128, seed=4, scale=0.2).requires_grad_(), rnd(8, 196, seed=5, scale=0.2).requires_grad_()]
    bs = [rnd(128, seed=6).requires_grad_(), None, rnd(8, seed=7).requires_grad_()]
    acts = [1, 2, 0]
    y = R.rows_mlp(x1, list(zip(Ws, bs)), acts, x2=x2)
    g = rnd(M, 8, seed=9)
    y.backward(g)
    d = [t.detach().double().requires_grad_() for t in (x1, x2, *Ws, bs[0], bs[2])]
    h = torch.cat([d[0], d[1]], 1)
    h = torch.relu(h @ d[2].t() + d[5])
    h = torch.tanh(h @ d[3].t())
    h = h @ d[4].t() + d[6]
    h.backward(g.double())
    close(y, h, what="forward")
    for got, want, name in zip((x1, x2, *Ws, bs[0], bs[2]), d, ("x1", "x2", "W0", "W1", "W2", "b0", "b2")):
        close(got.grad, want.grad, tol=2e-4, what=name)


def _plan_reference(idx, n, canon):
    S, M, ns = idx.shape
    rp, rg, off = [], [], [0]
    for s in range(S):
        for m in range(M):
            seen = []
            for j in range(ns):
                e = int(idx[s, m, j])
                if canon is not None:
                    e = int(canon[s, e])
                if e not in seen:
                    seen.append(e)
            rp += [s * n + e for e in seen]
            rg += [s * M + m] * len(seen)
            off.append(len(rp))
    return np.array(rp), np.array(rg), np.array(off)


def test_sa_rows_plan_distinct_entries_any_list_form():
    """ball-query form (distinct then copies of the first), cyclic copies under a canonical map, arbitrary lists with repeats
    anywhere, a single slot"""
    from jmodt_amd.ops import rows as R
    rng = np.random.default_rng(0)
    S, M, ns, n = 3, 37, 16, 50
    idx = rng.integers(0, n, size=(S, M, ns)).astype(np.int32)
    idx[0, :, 5:] = idx[0, :, :1]                        # ball-query back-fill
    idx[1, :, :] = (np.arange(ns)[None, :] % 7 + rng.integers(0, 20, size=(M, 1))).astype(np.int32)     # cyclic
    canon = np.arange(n, dtype=np.int32)[None].repeat(S, 0)
    canon[2] = canon[2] % 9
    for cn in (None, canon):
        plan = R.RowsPlan(torch.from_numpy(idx).to(DEV), n, torch.from_numpy(cn).to(DEV) if cn is not None else None)
        rp, rg, off = _plan_reference(idx, n, cn)
        total = int(plan.rows_dev.item())
        assert total == len(rp)
        assert np.array_equal(plan.offsets.cpu().numpy(), off)
        assert np.array_equal(plan.row_point.cpu().numpy()[:total], rp) and np.array_equal(plan.row_group.cpu().numpy()[:total], rg)
    one = R.RowsPlan(torch.zeros((2, 5, 1), dtype=torch.int32, device=DEV), 4)
    assert int(one.rows_dev.item()) == 10


@pytest.mark.parametrize("C,ns,group_all", [(0, 16, False), (32, 32, False), (16, 8, True)])
def test_sa_scale_rows_vs_the_operator_route(C, ns, group_all):
    """QueryAndGroup / GroupAll + SharedMLP + max-pool (pointnet2_modules.py:46-55) by torch autograd in float64 on the grouped
    tensors against the rows form: output, d(features), d(every weight)"""
    from jmodt_amd.ops import rows as R
    from jmodt_amd.ops.pointnet2 import pointnet2_utils as pu
    S, n, m = 3, 256, 32
    xyz = torch.from_numpy(synth.cloud(S, n, seed=11, dup_frac=0.1)).to(DEV)
    feats = rnd(S * n, C, seed=1).requires_grad_() if C else None
    widths = [3 + C, 32, 24, 40]
    Ws = [rnd(widths[i + 1], widths[i], seed=20 + i, scale=0.3).requires_grad_() for i in range(3)]
    bs = [rnd(widths[i + 1], seed=30 + i, scale=0.3).requires_grad_() for i in range(3)]
    if group_all:
        idx = torch.arange(n, dtype=torch.int32, device=DEV).expand(S, 1, n)[:, :, :ns].contiguous()
        ctr = None
        G = S
    else:
        _, new_xyz = pu.farthest_point_sample_xyz(xyz, m)
        idx = pu.ball_query(4.0, ns, xyz, new_xyz)
        ctr = new_xyz.reshape(-1, 3).contiguous()
        G = S * m
    plan = R.RowsPlan(idx, n)
    out = R.sa_scale_rows(feats, xyz.reshape(-1, 3), ctr, plan, list(zip(Ws, bs)))
    g = rnd(G, widths[-1], seed=40)
    out.backward(g)
    # float64 reference on the dense grouped rows
    li = idx.long()
    flat = (li + (torch.arange(S, device=DEV) * n)[:, None, None]).reshape(-1)
    d = [t.detach().double().requires_grad_() for t in Ws + bs]
    fd = feats.detach().double().requires_grad_() if C else None
    gx = xyz.reshape(-1, 3).double()[flat]
    if ctr is not None:
        gx = gx - ctr.double().repeat_interleave(ns, 0)
    x = torch.cat([gx, fd[flat]], 1) if C else gx
    for l in range(3):
        x = torch.relu(x @ d[l].t() + d[3 + l])
    want = x.view(G, ns, -1).amax(1)
    want.backward(g.double())
    close(out, want, what="pooled")
    for got, ref, name in zip(Ws + bs, d, ["W1", "W2", "W3", "b1", "b2", "b3"]):
        close(got.grad, ref.grad, tol=3e-4, what=name)
    if C:
        close(feats.grad, fd.grad, tol=3e-4, what="d features")


def test_three_interpolate_rows_and_feature_gather_rows_vs_operators():
    from jmodt_amd.ops import rows as R
    from jmodt_amd.ops.fusion import feature_gather
    from jmodt_amd.ops.pointnet2 import pointnet2_utils as pu
    B, n, m, C = 2, 500, 64, 48
    known = rnd(B * m, C, seed=1).requires_grad_()
    idx = torch.randint(0, m, (B, n, 3), generator=torch.Generator().manual_seed(2)).int().to(DEV)
    w = torch.rand(B, n, 3, generator=torch.Generator().manual_seed(3)).to(DEV)
    out = R.three_interpolate_rows(known, idx, w)
    g = rnd(B * n, C, seed=4)
    out.backward(g)
    k2 = known.detach().view(B, m, C).transpose(1, 2).contiguous().requires_grad_()
    ref = pu.three_interpolate(k2, idx, w)
    ref.backward(g.view(B, n, C).transpose(1, 2).contiguous())
    close(out.view(B, n, C).transpose(1, 2), ref, what="three_interpolate_rows")
    close(known.grad.view(B, m, C).transpose(1, 2), k2.grad, what="three_interpolate_rows grad")
    fmap = rnd(B, 16, 24, 40, seed=5).contiguous(memory_format=torch.channels_last).requires_grad_()
    xy = (torch.rand(B, n, 2, generator=torch.Generator().manual_seed(6)) * 2.2 - 1.1).to(DEV)
    got = R.feature_gather_rows(fmap, xy)
    g2 = rnd(B * n, 16, seed=7)
    got.backward(g2)
    f2 = fmap.detach().clone().requires_grad_()
    want = F.grid_sample(f2, xy.unsqueeze(1), mode="bilinear", padding_mode="zeros", align_corners=True).squeeze(2)
    want.backward(g2.view(B, n, 16).transpose(1, 2))
    close(got.view(B, n, 16).transpose(1, 2), want, what="feature_gather_rows")
    close(fmap.grad, f2.grad, what="feature_gather_rows grad")
    assert torch.equal(got.view(B, n, 16).transpose(1, 2).contiguous(), feature_gather(fmap.detach(), xy))


@pytest.mark.parametrize("cin,cout", [(3, 64), (64, 128), (128, 256)])
def test_image_block_first_layer_forward_kernels_and_winograd_data_gradient(cin, cout):
    """conv3x3 + bias + ReLU of the image blocks in the training path (train_rows._Conv3x3BiasRelu): one-pass forward kernels, data
    gradient of the Winograd layers = the same kernel on the flipped / transposed weight — against F.conv2d + relu under autograd"""
    from jmodt_amd.train_rows import _Conv3x3BiasRelu
    B, H, W = 2, 24, 40
    x = rnd(B, cin, H, W, seed=1)
    if cin != 3:
        x = x.contiguous(memory_format=torch.channels_last)
    x.requires_grad_(cin != 3)
    w = rnd(cout, cin, 3, 3, seed=2, scale=0.1).contiguous(memory_format=torch.channels_last).requires_grad_()
    b = rnd(cout, seed=3, scale=0.1).requires_grad_()
    y = _Conv3x3BiasRelu.apply(x, w, b)
    g = rnd(B, cout, H, W, seed=4).contiguous(memory_format=torch.channels_last)
    y.backward(g)
    x2 = x.detach().double().requires_grad_(cin != 3)
    w2, b2 = w.detach().double().requires_grad_(), b.detach().double().requires_grad_()
    pre = F.conv2d(x2, w2, b2, padding=1)
    close(y, F.relu(pre), what="forward")
    # the backward against float64 THROUGH THE SAME ReLU mask: an output whose pre-activation is within rounding of zero may sit on
    # the other side of the kink in float64, and one such element moves a weight gradient entry by |g x| ~ O(1)
    (pre * (y.detach() > 0).double()).backward(g.double())
    close(w.grad, w2.grad, tol=2e-4, what="d weight")
    close(b.grad, b2.grad, tol=2e-4, what="d bias")
    if cin != 3:
        close(x.grad, x2.grad, tol=2e-4, what="d input")


@pytest.fixture(scope="module")
def tiny():
    from jmodt_amd.detector import DetectorConfig
    from tests.test_gpu_detector import make_engine
    eng = make_engine(seed=3, cfg=DetectorConfig.tiny()).to(DEV).eval()
    xyz, img, xy = synth.frames(2, 2048, 77, H=96, W=320, native=(94, 310))
    xy = np.random.default_rng(5).uniform(-0.98, 0.98, size=xy.shape).astype(np.float32)
    for p in eng.parameters():
        p.requires_grad_(True)
    return eng, torch.from_numpy(xyz).to(DEV), torch.from_numpy(img).to(DEV), torch.from_numpy(xy).to(DEV)


def test_attention_rows_vs_the_module(tiny):
    from jmodt_amd.train_rows import BnFold, _attention_rows
    eng = tiny[0]
    mod = eng.rpn.backbone_net.Fusion_Conv[1]
    pc, ic = mod.conv1.in_channels // 2, mod.IA_Layer.fc1.in_features
    B, n = 2, 300
    p, i = rnd(B * n, pc, seed=1).requires_grad_(), rnd(B * n, ic, seed=2).requires_grad_()
    eng.zero_grad(set_to_none=True)
    out = _attention_rows(BnFold(eng), mod, p, i)
    g = rnd(B * n, out.shape[1], seed=3)
    out.backward(g)
    mine = {k: v.grad.clone() for k, v in mod.named_parameters()}
    dp, di = p.grad.clone(), i.grad.clone()
    eng.zero_grad(set_to_none=True)
    p2 = p.detach().view(B, n, pc).transpose(1, 2).contiguous().requires_grad_()
    i2 = i.detach().view(B, n, ic).transpose(1, 2).contiguous().requires_grad_()
    ref = mod(p2, i2)
    ref.backward(g.view(B, n, -1).transpose(1, 2).contiguous())
    close(out.view(B, n, -1).transpose(1, 2), ref, what="attention forward")
    close(dp.view(B, n, pc).transpose(1, 2), p2.grad, tol=3e-4, what="d point")
    close(di.view(B, n, ic).transpose(1, 2), i2.grad, tol=3e-4, what="d image")
    for k, v in mod.named_parameters():
        close(mine[k], v.grad, tol=3e-4, what=k)
    eng.zero_grad(set_to_none=True)


def _grads(eng):
    return {k: (None if v.grad is None else v.grad.detach().clone()) for k, v in eng.named_parameters()}


def _relative_gradient_error(mine, want):
    """worst per-tensor error relative to the tensor's largest entry; tensors whose gradient is analytically ~0 (the link head's last
    bias: the dual softmax is shift invariant) are measured against a floor of 1e-4 of the largest gradient in the network"""
    worst = ("", 0.0)
    gmax = max(float(w.abs().max()) for w in want.values() if w is not None)
    for k, w in want.items():
        assert (w is None) == (mine[k] is None), k
        if w is None:
            continue
        scale = max(float(w.abs().max()), 1e-4 * gmax)
        err = float((mine[k] - w).abs().max()) / scale
        if not err <= worst[1]:
            worst = (k, err)
    return worst, gmax


def test_joint_rows_route_matches_the_operator_route(tiny):
    """forward outputs and the gradient of EVERY parameter: the rows route (hand-written forward / backward kernels, BatchNorm
    folded, three streams, asynchronous) against the operator route (torch autograd over the grouped tensors), same engine, same
    frames, eval-mode BatchNorm.  The operator route's autograd graph stays referenced (`ref`) while the rows route runs — the way a
    training loop keeps last step's outputs: round 5's device fault under exactly that was MIOpen reading past a
    512-byte folded weight at the end of an allocator segment, cured in ops/rows._slab_like / train_rows._image_fusion_map"""
    from jmodt_amd import train_joint
    from jmodt_amd.train_rows import joint_forward_rows
    eng, xyz, img, xy = tiny
    K = eng.cfg.rpn_post_nms_top_n
    tids = torch.randint(0, 6, (2, K), generator=torch.Generator().manual_seed(4)).float().to(DEV)
    eng.zero_grad(set_to_none=True)
    ref = train_joint.joint_forward(eng, xyz, img, xy, rois_per_frame=K)
    train_joint.thin_loss(eng, ref, tids).backward()
    want = _grads(eng)
    eng.zero_grad(set_to_none=True)
    got = joint_forward_rows(eng, xyz, img, xy, rois_per_frame=K)
    train_joint.thin_loss(eng, got, tids).backward()
    mine = _grads(eng)
    eng.zero_grad(set_to_none=True)
    for k in ("backbone_features", "rpn_cls", "rpn_reg"):
        close(got[k], ref[k], what=k)
    assert torch.equal(got["rois"], ref["rois"]) or float((got["rois"] - ref["rois"]).abs().max()) < 1e-3
    for k in ("rcnn_cls", "rcnn_reg", "rcnn_feat"):
        close(got[k], ref[k], tol=2e-4, what=k)
    worst, gmax = _relative_gradient_error(mine, want)
    print("worst relative gradient error", worst, "largest gradient", gmax)
    assert worst[1] < 5e-4, worst


def test_rows_backward_under_held_graphs_ten_asynchronous_steps(tiny):
    """the pattern torch itself warns about (DDP stashes AccumulateGrad nodes; a loop keeps last step's loss dict): the operator route's
    graph AND the previous rows step's graph stay referenced while the next rows step runs its asynchronous three-stream forward and
    backward, ten times, no synchronisation in between — every step's gradients equal the ones of a step run with a device
    synchronisation behind every library call"""
    from jmodt_amd import _lib as L_
    from jmodt_amd import train_joint
    from jmodt_amd.train_rows import joint_forward_rows
    eng, xyz, img, xy = tiny
    K = eng.cfg.rpn_post_nms_top_n
    tids = torch.randint(0, 6, (2, K), generator=torch.Generator().manual_seed(4)).float().to(DEV)
    eng.zero_grad(set_to_none=True)
    stale = train_joint.joint_forward(eng, xyz, img, xy, rois_per_frame=K)       # its AccumulateGrad nodes live on the main stream
    train_joint.thin_loss(eng, stale, tids).backward()
    eng.zero_grad(set_to_none=True)
    L_.SYNC_DEBUG = True
    try:
        g = joint_forward_rows(eng, xyz, img, xy, rois_per_frame=K)
        train_joint.thin_loss(eng, g, tids).backward()
        torch.cuda.synchronize()
    finally:
        L_.SYNC_DEBUG = False
    want = _grads(eng)
    eng.zero_grad(set_to_none=True)
    kept, snaps = g, []
    for _ in range(10):
        g = joint_forward_rows(eng, xyz, img, xy, rois_per_frame=K)
        train_joint.thin_loss(eng, g, tids).backward()
        kept = g                                  # the previous step's graph dies only now, under this step's queued kernels
        snaps.append(_grads(eng))
        eng.zero_grad(set_to_none=True)
    torch.cuda.synchronize()
    assert stale["rcnn_feat"].grad_fn is not None and kept["rcnn_feat"].grad_fn is not None
    for i, mine in enumerate(snaps):
        worst, _ = _relative_gradient_error(mine, want)
        assert worst[1] < 1e-4, (i, worst)


def test_folded_weights_come_from_one_slab_with_slack_behind_the_last(tiny):
    """no folded weight can be the last bytes of an allocator segment (a convolution library that over-reads a small weight — MIOpen's
    1 x 1 data gradient does, tools/miopen_oob_probe.py — stays inside the fold's own allocation)"""
    from jmodt_amd.ops import rows as R
    from jmodt_amd.train_rows import BnFold, bn_pairs
    eng = tiny[0]
    fold = BnFold(eng)
    ws = [fold._slots[id(conv)][0] for conv, _ in bn_pairs(eng)]
    base = ws[0].untyped_storage()
    end = base.data_ptr() + base.nbytes()
    for (conv, _), w in zip(bn_pairs(eng), ws):
        assert w.untyped_storage().data_ptr() == base.data_ptr()
        assert w.data_ptr() % 256 == 0 and w.stride() == conv.weight.stride()
        assert w.data_ptr() + 4 * w.numel() + R.FOLD_SLAB_SLACK <= end


def test_joint_step_rows_route_updates_every_parameter(tiny):
    from jmodt_amd import train_joint
    from jmodt_amd.detector import DetectorConfig
    from tests.test_gpu_detector import make_engine
    _, xyz, img, xy = tiny
    eng = make_engine(seed=3, cfg=DetectorConfig.tiny()).to(DEV)
    train_joint.freeze_bn(eng)
    for m in eng.modules():
        if isinstance(m, torch.nn.Dropout):
            m.eval()
    for p in eng.parameters():
        p.requires_grad_(True)
    params = list(eng.parameters())
    opt = torch.optim.Adam(params, lr=1e-3, fused=True)
    K = min(64, eng.cfg.rpn_post_nms_top_n)
    tids = torch.randint(0, 6, (2, K), generator=torch.Generator().manual_seed(4)).float().to(DEV)
    before = [p.detach().clone() for p in params]
    loss = train_joint.joint_step(eng, xyz, img, xy, tids, opt, rois_per_frame=K, route="rows", next_xyz=xyz)
    loss2 = train_joint.joint_step(eng, xyz, img, xy, tids, opt, rois_per_frame=K, route="rows")     # takes the announced pyramid
    torch.cuda.synchronize()
    assert torch.isfinite(loss) and torch.isfinite(loss2)
    assert not [n for n, p in eng.named_parameters() if p.grad is None]
    assert all(bool(torch.isfinite(p.grad).all()) for p in params)
    moved = sum(int(not torch.equal(a, b)) for a, b in zip(before, params))
    assert moved >= len(params) - 12, (moved, len(params))


def test_rcnn_rows_on_canonical_centres_same_outputs_same_gradients_fewer_rows(tiny, monkeypatch):
    """RoI sets full of cyclic copies (roipool3d_kernel.cu:123-160): planning levels 2 and 3 on the first of every class of copied
    centres leaves the forward bit-identical (copies have bit-identical features, max-pool is idempotent) and moves each copy's
    gradient to its representative — the parameters' gradients agree to rounding (summation order)"""
    from jmodt_amd import train_rows as TR
    from jmodt_amd.ops import rows as R
    eng = tiny[0]
    S = eng.cfg.rcnn_num_points
    Rn = 12
    k = eng.rcnn_net.rcnn_input_channel
    Cf = eng.rcnn_net.merge_down_layer[0].conv.in_channels - eng.rcnn_net.xyz_up_layer[-1].conv.out_channels
    g = torch.Generator().manual_seed(11)
    base = torch.randn(Rn, S, k + Cf, generator=g)
    base[:, :, :3] *= 0.6
    count = torch.tensor([1, 3, 7, 20, 33, 64, 100, S, 5, 2, 50, 17], dtype=torch.int32)[:Rn].clamp(max=S)
    slot = torch.arange(S).view(1, S) % count.view(-1, 1).long()
    pts = torch.gather(base, 1, slot.unsqueeze(-1).expand(-1, -1, k + Cf)).to(DEV)                 # rows count .. S - 1 are cyclic copies
    count = count.to(DEV)
    seen = []
    plan_init = R.RowsPlan.__init__

    def counting(self, idx, n, canon=None):
        plan_init(self, idx, n, canon)
        seen[-1].append(self.rows_dev)
    monkeypatch.setattr(R.RowsPlan, "__init__", counting)
    res = {}
    for on in (False, True):
        monkeypatch.setattr(TR, "CANON_CENTRES", on)
        seen.append([])
        eng.zero_grad(set_to_none=True)
        out = TR.rcnn_forward_rows(eng, pts, TR.BnFold(eng), count)
        w = rnd(*out["rcnn_feat"].shape, seed=5)
        ((out["rcnn_feat"] * w).sum() + out["rcnn_cls"].sum() + (out["rcnn_reg"] * rnd(*out["rcnn_reg"].shape, seed=6)).sum()).backward()
        res[on] = ({kk: v.detach().clone() for kk, v in out.items()},
                   {kk: v.grad.detach().clone() for kk, v in eng.rcnn_net.named_parameters() if v.grad is not None})
    eng.zero_grad(set_to_none=True)
    rows = [[int(r) for r in s_] for s_ in seen]
    print("rows per level, copies kept / canonical centres:", rows)
    assert rows[0][0] == rows[1][0] and all(a >= b for a, b in zip(rows[0], rows[1])) and sum(rows[1]) < sum(rows[0])
    for kk in res[False][0]:
        assert torch.equal(res[False][0][kk], res[True][0][kk]), kk
    assert res[False][1].keys() == res[True][1].keys() and len(res[True][1]) > 10
    for kk, want in res[False][1].items():
        close(res[True][1][kk], want, tol=2e-4, what=kk)


@pytest.mark.parametrize("M,n,k", [(70000, 16, 16), (70000, 32, 16), (65537, 64, 32), (40001, 32, 64), (300000, 64, 64), (9000, 128, 64), (9001, 64, 128),
                                   (50000, 128, 128), (3000, 512, 256), (2, 16, 4), (511, 36, 132), (100000, 16, 4), (1200, 256, 260)])
def test_rows_wgrad_direct_form_every_patch_shape(M, n, k):
    """the weight-gradient kernel's wave patches (32 / 64 columns per operand, 1 / 2 / 4 waves per tile, spare waves on other rows of
    the split), split and un-split, odd row counts, the row count in device memory, against float64; run to run bit-identical"""
    from jmodt_amd.ops import rows as R
    dy, x = rnd(M, n, seed=1), rnd(M, k, seed=2)
    dw, db = R.linear_wgrad(dy, [x])
    scale = max(1.0, float(M) ** 0.5 / 30)
    close(dw, dy.double().t() @ x.double(), tol=2e-4 * scale, what="wgrad")
    close(db, dy.double().sum(0), tol=2e-4 * scale, what="bias gradient")
    for mv in (0, 1, M // 3 + 1, M - 1):
        if mv > M:
            continue
        m_dev = torch.tensor([mv], dtype=torch.int32, device=DEV)
        dw2, db2 = R.linear_wgrad(dy, [x], m_dev=m_dev)
        close(dw2, dy[:mv].double().t() @ x[:mv].double(), tol=2e-4 * scale, what=f"wgrad, {mv} rows on the device")
        close(db2, dy[:mv].double().sum(0), tol=2e-4 * scale, what=f"bias gradient, {mv} rows on the device")
        again = R.linear_wgrad(dy, [x], m_dev=m_dev)
        assert torch.equal(dw2, again[0]) and torch.equal(db2, again[1])
