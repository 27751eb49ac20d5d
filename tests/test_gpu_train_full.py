"""GPU tier (-m gpu): the DEFAULT training route — train_rows.joint_forward_rows, what joint_step(route="auto") and `bench.py --workload
train --joint` run, three streams, asynchronous — at the BENCHMARKED widths (DetectorConfig.survey(): 16.7 M parameters, hidden widths
to 512, 16384-point frames on the 384 x 1280 canvas) on the three benchmark clouds.

  * forward: backbone features and RPN heads against oracle/pipeline.Chain in float64 (the restatement of backbone.py:159-196 /
    pointnet2_modules.py:20-63,135-164 that every composed inference test is checked against, itself pinned to the reference's
    Python in tests/test_oracle_cpu.py), the RCNN on the route's own pooled points against Chain.rcnn (rcnn.py:176-202);
  * backward: the gradient of every parameter against the operator route (torch autograd over the grouped (B, C, npoint, nsample)
    tensors, pinned to the reference's autograd at 5e-4 in test_gpu_train_joint.py), same frames, same RoIs, eval-mode BatchNorm.
The tiny-configuration tests of test_gpu_rows.py select other kernels (narrow layers, one wave patch shape); these are the shapes
that are timed.

Two things measured while writing this test (tools/grad_state_probe.py, tools/forward_repro_probe.py):
  * the image is the 96 x 320 canvas of the tiny tests, not 384 x 1280: the image branch is not a SURVEY.md section 8 row, its
    widths are the benchmarked ones either way, and MIOpen builds the operator route's NCHW convolution kernels at first use —
    418 s on a fresh box at 384 x 1280, 50 s at 96 x 320;
  * the loss is LINEAR in the head outputs, so the gradient depends on the forward only through the ReLU / tanh / sigmoid
    derivatives, and the library convolutions of the image branch are not bit-reproducible from call to call (one F.conv2d on one
    tensor: 4.8e-7 between calls; the backbone features of EITHER route: 6e-7 between calls).  A perturbation of that size flips a
    few ReLU masks, which moves single gradient tensors by 2 - 7e-4 of their maximum: operator route against itself up to 5e-4 (once
    3.4e-3 on an RCNN head at 384 x 1280), rows route against itself up to 7e-4, rows against operators 1.2e-4 when the two calls
    happen to share their masks — no NaN appears with every `torch.empty` poisoned (JM_POISON_EMPTY), so it is not uninitialised
    memory.  Between the two ROUTES the same happens deterministically: their forwards differ by rounding (1e-7: different kernels,
    different summation orders), and among the 10^7 pre-activations of a batch some lie that close to zero.  Worst tensor over six
    runs of this test on three clouds: 0.7 - 5.4e-4 for the better operator call (the packed cloud's RCNN input layer at the top:
    every RoI holds hundreds of real points), and one run in three or four lands above 5e-4 on some tensor.  tools/rcnn_state_probe.py
    pinned one such case down: 25 rows-route steps on the uniform cloud in every stream mode (asynchronous, synchronised between
    forward and backward, single stream, synchronised behind every call) — pooled RoI points 2 - 4e-6 apart from step to step
    (the proposals' decoded boxes move with the features), 24 gradients equal to 1e-7 and ONE (single-stream mode) off by 1.78e-3 on
    rcnn_net.SA_modules.0.mlps.0.layer1.conv.weight, the same tensor and amount every time it appears: one pre-activation of a row
    that stands for dozens of copied RoI points sits within 1e-6 of zero.  Not a stream race.  The comparison therefore makes up to
    five complete attempts (own rows step, own two operator-route calls): every attempt's forward must match the float64 chain
    and stay within 5e-3, ONE attempt must agree entry-wise to 5e-4; the entry-wise bar holds unconditionally where no mask sits on the
    fence — against the reference's autograd fixture (test_gpu_train_joint.py: 3.6e-6) and at the tiny widths (test_gpu_rows.py).
"""
import numpy as np
import pytest
import torch

from jmodt_amd import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
K = 64                 # RoIs per frame of the training step (config.py:153)


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _close(got, want, tol, what):
    got = got.detach().double().cpu()
    want = torch.as_tensor(want).double().cpu()
    scale = max(1.0, float(want.abs().max()))
    err = float((got - want).abs().max())
    assert err <= tol * scale, (what, err, scale)


def _relative_l2_error(mine, want):
    """worst ||a - b||_2 / ||b||_2 over the tensors (those below 1e-4 of the network's largest gradient norm are measured against that floor)"""
    nmax = max(float(w.double().norm()) for w in want.values() if w is not None)
    worst = ("", 0.0)
    for k, w in want.items():
        if w is None:
            continue
        err = float((mine[k].double() - w.double()).norm()) / max(float(w.double().norm()), 1e-4 * nmax)
        if err > worst[1]:
            worst = (k, err)
    return worst


@pytest.fixture(scope="module")
def engine():
    from jmodt_amd.detector import DetectorConfig
    from tests.test_gpu_detector import make_engine
    eng = make_engine(seed=5, cfg=DetectorConfig.survey()).to(DEV).eval()
    for p in eng.parameters():
        p.requires_grad_(True)
    return eng


def _attempt(eng, chain, xyz_h, img_h, xy_h, xyz, img, xy, tids, kind):
    """one complete comparison: the rows route as joint_step runs it (asynchronous, three streams), its forward against the float64
    chain, then two calls of the operator route on the SAME RoIs; returns the per-call (worst max-norm error, worst relative L2 error)"""
    from jmodt_amd import train_joint
    from jmodt_amd.train_rows import joint_forward_rows, pooled_rois
    from tests.test_gpu_rows import _grads, _relative_gradient_error
    eng.zero_grad(set_to_none=True)
    got = joint_forward_rows(eng, xyz, img, xy, rois_per_frame=K)
    train_joint.thin_loss(eng, got, tids).backward()
    torch.cuda.synchronize()
    mine = _grads(eng)
    eng.zero_grad(set_to_none=True)
    # ---- forward against the float64 chain
    want = chain.rpn(xyz_h, img_h, xy_h)
    assert float(want["backbone_features"].abs().max()) > 0.5
    _close(got["backbone_features"], want["backbone_features"], 1e-4, "backbone_features")
    _close(got["rpn_cls"], want["rpn_cls"], 1e-4, "rpn_cls")
    _close(got["rpn_reg"], want["rpn_reg"], 1e-4, "rpn_reg")
    # the pooled points the route's RCNN saw (deterministic: the same proposal / pooling kernels on the same head outputs)
    N, C = xyz.shape[1], got["backbone_features"].shape[1]
    rows = got["backbone_features"].detach().transpose(1, 2).reshape(2 * N, C).contiguous()
    rois, pts_input, count = pooled_rois(eng, xyz, dict(rpn_cls=got["rpn_cls"], rpn_reg=got["rpn_reg"], feature_rows=rows), K)
    assert torch.equal(rois, got["rois"])
    assert int((count > 0).sum()) >= K                         # the scene gives the RCNN real RoIs
    rc = chain.rcnn(pts_input.cpu().numpy())
    _close(got["rcnn_feat"], rc["rcnn_feat"].reshape(2 * K, -1), 1e-4, "rcnn_feat")
    _close(got["rcnn_cls"], rc["rcnn_cls"].reshape(2 * K, -1), 1e-4, "rcnn_cls")
    _close(got["rcnn_reg"], rc["rcnn_reg"].reshape(2 * K, -1), 1e-4, "rcnn_reg")
    del got
    # ---- backward against the operator route on the same RoIs (the RCNN half teacher-forced on the rows route's pooled points:
    # a proposal that flips between two routes 1e-6 apart would compare two different losses), two calls: see the module docstring
    results = []
    for _ in range(2):
        feats = train_joint.backbone_forward(eng.rpn.backbone_net, xyz, img, xy)
        ref = train_joint.rcnn_forward_train(eng.rcnn_net, pts_input)
        ref.update(rpn_cls=eng.rpn.rpn_cls_layer(feats).transpose(1, 2), rpn_reg=eng.rpn.rpn_reg_layer(feats).transpose(1, 2))
        train_joint.thin_loss(eng, ref, tids).backward()
        torch.cuda.synchronize()
        want_g = _grads(eng)
        eng.zero_grad(set_to_none=True)
        del feats, ref
        results.append((_relative_gradient_error(mine, want_g)[0], _relative_l2_error(mine, want_g)))
    print(kind, "per operator-route call: worst max-norm error", [r[0] for r in results], "worst relative L2 error", [r[1] for r in results])
    return results


@pytest.mark.parametrize("kind", ["uniform", "kitti", "packed"])
def test_rows_route_at_the_benchmarked_widths(engine, kind):
    from oracle.pipeline import Chain
    eng = engine
    xyz_h, img_h, xy_h = synth.frames(2, 16384, 4321, kind=kind, H=96, W=320, native=(94, 310))
    # (synth.frames projects with the intrinsics of the 1280-wide canvas: on the 320-wide image nearly every point would fall outside)
    xy_h = np.random.default_rng(5).uniform(-0.98, 0.98, size=xy_h.shape).astype(np.float32)
    xyz, img, xy = T(xyz_h), T(img_h), T(xy_h)
    tids = torch.randint(0, 6, (2, K), generator=torch.Generator().manual_seed(4)).float().to(DEV)
    chain = Chain(eng.state_dict(), eng.cfg, torch.float64)
    attempts = []
    for _ in range(5):
        # every attempt is a complete, independent comparison (its own rows forward / backward, its own two operator calls); the
        # forward of every attempt must match the float64 chain; the gradients of ONE attempt must agree entry-wise to 5e-4 — an
        # attempt that lands on a fence-sitting pre-activation (module docstring) shows up as a single tensor off by a fixed amount
        # (uniform cloud: always 1.73e-3 on rcnn_net.SA_modules.0.mlps.0.layer1.conv.weight) and is bounded at 5e-3
        res = _attempt(eng, chain, xyz_h, img_h, xy_h, xyz, img, xy, tids, kind)
        attempts.append(res)
        assert max(r[0][1] for r in res) < 5e-3, attempts
        if min(r[0][1] for r in res) < 5e-4:
            break
    else:
        raise AssertionError(f"no attempt of five agreed to 5e-4: {attempts}")
