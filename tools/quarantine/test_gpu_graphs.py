"""HIP-graph sections (jmodt_amd/graphed.py) and the joint-mode step on them (jmodt_amd/train_graphs.py): a replayed section equals
the eager function — outputs, input gradients, parameter gradients — step after step with changing inputs, and the 'graphs' route
of the joint step equals the 'rows' route (same kernels, so the bar is the rows route's own run-to-run band)."""
import gc

import numpy as np
import pytest
import torch

from jmodt_amd import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def close(got, want, tol=1e-5, what=""):
    got, want = got.detach().double().cpu(), want.detach().double().cpu()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    scale = max(1.0, float(want.abs().max()))
    err = float((got - want).abs().max())
    assert err <= tol * scale, (what, err, scale)


def test_section_replay_equals_eager_forward_and_backward():
    from jmodt_amd.graphed import GraphedSection
    from jmodt_amd.ops import rows as R
    g = torch.Generator().manual_seed(0)
    W1 = torch.nn.Parameter((torch.randn(64, 32, generator=g) * 0.2).to(DEV))
    b1 = torch.nn.Parameter(torch.randn(64, generator=g).to(DEV) * 0.1)
    W2 = torch.nn.Parameter((torch.randn(16, 64, generator=g) * 0.2).to(DEV))

    def fn(x, scale, w1, bb1, w2):
        h = R.rows_mlp(x, [(w1, bb1), (w2, None)], [1, 0])            # hand-written forward / backward kernels inside
        return h * scale, (h > 0).sum(dim=1).int()                     # + a torch op, + a non-differentiable output

    sec = GraphedSection(fn, "toy")
    for step in range(4):
        x = torch.randn(300, 32, generator=g).to(DEV).requires_grad_(True)
        scale = torch.rand(300, 1, generator=g).to(DEV) + 0.5
        go = torch.randn(300, 16, generator=g).to(DEV)
        for p in (W1, b1, W2):
            p.grad = None
        y, cnt = sec(x, scale, W1, b1, W2)
        y_keep, cnt_keep = y.detach().clone(), cnt.clone()
        y.backward(go)
        got = [x.grad.clone(), W1.grad.clone(), b1.grad.clone(), W2.grad.clone()]
        x2 = x.detach().clone().requires_grad_(True)
        for p in (W1, b1, W2):
            p.grad = None
        y2, cnt2 = fn(x2, scale, W1, b1, W2)
        y2.backward(go)
        want = [x2.grad, W1.grad, b1.grad, W2.grad]
        close(y_keep, y2, what=f"step {step} forward")
        assert torch.equal(cnt_keep, cnt2)
        for a, b, n in zip(got, want, ("dx", "dW1", "db1", "dW2")):
            close(a, b, what=f"step {step} {n}")
    assert sec.captures == 1 and sec.entries() == 1
    # a second backward pass without zero_grad accumulates (the parameter's .grad IS the section's buffer after the first)
    for p in (W1, b1, W2):
        p.grad = None
    x = torch.randn(300, 32, generator=g).to(DEV).requires_grad_(True)
    scale = torch.ones(300, 1, device=DEV)
    y, _ = sec(x, scale, W1, b1, W2)
    y.sum().backward()
    first = W1.grad.clone()
    y, _ = sec(x, scale, W1, b1, W2)
    y.sum().backward()
    close(W1.grad, 2 * first, what="accumulated over two passes")
    # without grad: forward graph only, same numbers
    with torch.no_grad():
        y3, _ = sec(x.detach(), scale, W1, b1, W2)
        want3, _ = fn(x.detach(), scale, W1, b1, W2)
    close(y3, want3, what="no-grad entry")
    assert sec.entries() == 2


def test_section_keys_static_inputs_by_address():
    from jmodt_amd.graphed import GraphedSection, mark_static
    sec = GraphedSection(lambda a, b: a * 2 + b, "addr")
    a0, a1 = mark_static(torch.ones(64, device=DEV)), mark_static(torch.full((64,), 3.0, device=DEV))
    b = torch.arange(64, device=DEV, dtype=torch.float32)
    assert torch.equal(sec(a0, b), a0 * 2 + b)
    assert torch.equal(sec(a1, b), a1 * 2 + b)
    a0.fill_(5.0)                                        # a static input is read in place
    assert torch.equal(sec(a0, b + 1), a0 * 2 + b + 1)    # a plain input is copied in at every call
    assert sec.entries() == 2 and sec.captures == 2


@pytest.fixture(scope="module")
def tiny():
    from jmodt_amd.detector import DetectorConfig
    from jmodt_amd import train_joint
    from tests.test_gpu_detector import make_engine
    eng = make_engine(seed=3, cfg=DetectorConfig.tiny()).to(DEV)
    train_joint.prepare_rows(eng)
    for m in eng.modules():
        if isinstance(m, torch.nn.Dropout):
            m.eval()
    xyz, img, xy = synth.frames(2, 2048, 77, H=96, W=320, native=(94, 310))
    xy = np.random.default_rng(5).uniform(-0.98, 0.98, size=xy.shape).astype(np.float32)
    for p in eng.parameters():
        p.requires_grad_(True)
    return eng, torch.from_numpy(xyz).to(DEV), torch.from_numpy(img).to(DEV), torch.from_numpy(xy).to(DEV)


def _grads(eng):
    return {k: (None if v.grad is None else v.grad.detach().clone()) for k, v in eng.named_parameters()}


def test_joint_graphs_route_matches_the_rows_route(tiny):
    """forward outputs and the gradient of EVERY parameter, three steps in a row (capture, replay, replay) on two different batches"""
    from jmodt_amd import train_graphs, train_joint
    from jmodt_amd.train_rows import joint_forward_rows
    eng, xyz, img, xy = tiny
    K = min(64, eng.cfg.rpn_post_nms_top_n)
    B = xyz.shape[0]
    batches = [(xyz, img, xy), (xyz.flip(0).contiguous(), img.flip(0).contiguous(), xy.flip(0).contiguous()), (xyz, img, xy)]
    for step, (x, im, pxy) in enumerate(batches):
        tids = torch.randint(0, 6, (B, K), generator=torch.Generator().manual_seed(4 + step)).float().to(DEV)
        eng.zero_grad(set_to_none=True)
        ref = joint_forward_rows(eng, x, im, pxy, rois_per_frame=K)
        train_joint.thin_loss(eng, ref, tids).backward()
        torch.cuda.synchronize()
        want = _grads(eng)
        # nothing of the eager step's autograd graph may stay alive: its AccumulateGrad nodes remember the side streams they were
        # made on, and a parameter whose stale accumulator lives on another stream pulls that stream into a section's capture
        ref = {k: v.detach() for k, v in ref.items()}
        gc.collect()
        eng.zero_grad(set_to_none=True)
        loss, got = train_graphs.forward_backward(eng, x, im, pxy, tids, None, True, K, None)
        torch.cuda.synchronize()
        mine = _grads(eng)
        eng.zero_grad(set_to_none=True)
        for k in ("rpn_cls", "rpn_reg", "rcnn_cls", "rcnn_reg", "rcnn_feat"):
            close(got[k].reshape(ref[k].shape), ref[k], tol=2e-5, what=f"step {step} {k}")
        close(got["backbone_features"].view(B, -1, got["backbone_features"].shape[-1]).transpose(1, 2), ref["backbone_features"], tol=2e-5,
              what="features")
        assert torch.equal(got["rois"], ref["rois"])
        gmax = max(float(w.abs().max()) for w in want.values() if w is not None)
        worst = ("", 0.0)
        for k, w in want.items():
            assert (w is None) == (mine[k] is None), (step, k)
            if w is None:
                continue
            scale = max(float(w.abs().max()), 1e-4 * gmax)
            err = float((mine[k] - w).abs().max()) / scale
            if err > worst[1]:
                worst = (k, err)
        print("step", step, "worst relative gradient difference graphs vs rows", worst)
        assert worst[1] < 2e-4, (step, worst)
    caps = train_graphs.joint_graphs(eng).captures()
    assert all(v == 1 for v in caps.values()), caps            # everything was captured once and replayed afterwards


def test_joint_step_graphs_route_trains_with_the_prefetched_geometry(tiny):
    from jmodt_amd import train_graphs, train_joint
    from jmodt_amd.detector import DetectorConfig
    from tests.test_gpu_detector import make_engine
    _, xyz, img, xy = tiny

    def run(route):
        eng = make_engine(seed=3, cfg=DetectorConfig.tiny()).to(DEV)
        train_joint.prepare_rows(eng)
        for m in eng.modules():
            if isinstance(m, torch.nn.Dropout):
                m.eval()
        for p in eng.parameters():
            p.requires_grad_(True)
        params = list(eng.parameters())
        opt = torch.optim.Adam(params, lr=1e-3, fused=True)
        K = min(64, eng.cfg.rpn_post_nms_top_n)
        tids = torch.randint(0, 6, (2, K), generator=torch.Generator().manual_seed(4)).float().to(DEV)
        losses = []
        for _ in range(3):
            losses.append(train_joint.joint_step(eng, xyz, img, xy, tids, opt, rois_per_frame=K, route=route, next_xyz=xyz, local=True))
        torch.cuda.synchronize()
        assert not [n for n, p in eng.named_parameters() if p.grad is None]
        return [float(l) for l in losses], [p.detach().clone() for p in params], eng

    l_rows, p_rows, _ = run("rows")
    l_graphs, p_graphs, eng = run("graphs")
    assert all(np.isfinite(l_graphs))
    # three Adam steps on the same batches: the trajectories agree (Adam's normalisation amplifies rounding differences of
    # near-zero gradients, hence the loose bar on the parameters and the tight one on the losses)
    for a, b in zip(l_rows, l_graphs):
        assert abs(a - b) <= 2e-3 * max(1.0, abs(a)), (l_rows, l_graphs)
    moved = sum(int(not torch.equal(a, b)) for a, b in zip(p_rows, p_graphs))
    print("losses rows", l_rows, "graphs", l_graphs, "parameter tensors that differ in some bit", moved, "of", len(p_rows))
    caps = train_graphs.joint_graphs(eng).captures()
    assert all(v == 1 for v in caps.values()), caps
