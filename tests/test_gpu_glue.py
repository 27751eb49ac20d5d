"""The single-launch glue passes of round 5 (csrc/detections.hip, csrc/elementwise.hip, csrc/points_gemm.hip) against the framework
formulations they replace (which restate the reference: tools/eval.py:171-193, pointnet2_modules.py:147-150, backbone.py:170-171,
point_rcnn.py:42-44 / proposal_target_layer.py:26, pytorch_utils.py:6-33)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _select_reference(pred_boxes3d, raw_scores, feats, score_thresh, nms_thresh):
    """round 2's formulation of select_detections: ~25 framework launches around jm_nms_batched"""
    from jmodt_amd.ext import iou3d_cuda
    from jmodt_amd.ops.iou3d.iou3d_utils import boxes3d_to_bev_torch
    B, M = raw_scores.shape
    dev = raw_scores.device
    norm = torch.sigmoid(raw_scores)
    valid = norm > score_thresh
    key = torch.where(valid, raw_scores, raw_scores.new_full((), float("-inf")))
    order = torch.sort(key, dim=1, descending=True, stable=True)[1]
    counts = valid.sum(dim=1).to(torch.int32)
    sorted_boxes = torch.gather(pred_boxes3d, 1, order.unsqueeze(-1).expand(-1, -1, 7))
    bev = boxes3d_to_bev_torch(sorted_boxes.view(-1, 7)).view(B, M, 5).contiguous()
    keep, num_keep = iou3d_cuda.nms_batched_device(bev, counts, nms_thresh, 0)
    slot = torch.arange(M, device=dev).unsqueeze(0)
    live = slot < num_keep.unsqueeze(1)
    keep = torch.where(live, keep, torch.zeros_like(keep))
    src = torch.gather(order, 1, keep)
    zero = pred_boxes3d.new_zeros(())
    boxes = torch.where(live.unsqueeze(-1), torch.gather(pred_boxes3d, 1, src.unsqueeze(-1).expand(-1, -1, 7)), zero)
    out_feats = torch.where(live.unsqueeze(-1), torch.gather(feats, 1, src.unsqueeze(-1).expand(-1, -1, feats.shape[2])), zero)
    return dict(boxes=boxes, scores=torch.where(live, torch.gather(norm, 1, src), zero), raw_scores=torch.where(live, torch.gather(raw_scores, 1, src), zero),
                feats=out_feats, count=num_keep.to(torch.int32), roi_index=torch.where(live, src, torch.zeros_like(src)))


@pytest.mark.parametrize("B,M,C,seed", [(8, 128, 512, 0), (3, 100, 64, 1), (1, 1, 8, 2), (2, 37, 12, 3),
                                        (2, 1500, 16, 4), (1, 20000, 4, 5)])          # (> 1024 / > 16384 slots: keys in dynamic LDS)
def test_select_detections_two_kernels_equal_the_framework_formulation(B, M, C, seed):
    from jmodt_amd.ops.detections import select_detections
    g = torch.Generator().manual_seed(seed)
    spread = 8.0 * max(1.0, (M / 128.0) ** 0.5)
    centre = torch.rand(B, M, 3, generator=g) * torch.tensor([spread, 1.0, spread])    # crowded: the NMS has work to do
    size = torch.tensor([1.5, 1.6, 3.9]) * (0.9 + 0.2 * torch.rand(B, M, 3, generator=g))
    ry = (torch.rand(B, M, 1, generator=g) * 2 - 1) * 3.14159
    boxes = torch.cat([centre, size, ry], dim=2).to(DEV)
    raw = (torch.randn(B, M, generator=g) * 2).to(DEV)
    if M > 1:
        raw[:, ::7] = raw[:, 1:2]                                                         # ties: the sort must be stable
    if M > 4:
        raw[0, :4] = -30.0                                                               # rejected slots in front
    feats = torch.randn(B, M, C, generator=g).to(DEV)
    got = select_detections(boxes, raw, feats, 0.2, 0.1)
    want = _select_reference(boxes, raw, feats, 0.2, 0.1)
    for k, w in want.items():
        assert torch.equal(getattr(got, k), w), k
    # nothing accepted at all
    got = select_detections(boxes, torch.full_like(raw, -20.0), feats, 0.2, 0.1)
    assert int(got.count.sum()) == 0 and float(got.boxes.abs().sum()) == 0.0 and float(got.feats.abs().sum()) == 0.0


def test_three_nn_weights_and_point_row_gather():
    from jmodt_amd.ops.pointnet2 import pointnet2_utils as pu
    g = torch.Generator().manual_seed(0)
    unknown = (torch.rand(3, 1000, 3, generator=g) * 10).to(DEV)
    known = unknown[:, ::5].contiguous() + 0.01                                          # includes near-coincident points
    known[:, 0] = unknown[:, 0]                                                          # and an exact hit (distance 0)
    idx, w = pu.three_nn_weights(unknown, known)
    d, idx0 = pu.three_nn(unknown, known)
    inv = 1.0 / (d.double() + 1e-8)
    want = inv / inv.sum(dim=2, keepdim=True)
    assert torch.equal(idx, idx0)
    assert float((w.double() - want).abs().max()) < 1e-6
    src = torch.randn(3, 1000, 2, generator=g).to(DEV)
    pick = torch.randint(0, 1000, (3, 77), generator=g).int().to(DEV)
    assert torch.equal(pu.gather_point_rows(src, pick), torch.gather(src, 1, pick.long().unsqueeze(-1).expand(-1, -1, 2)))


@pytest.mark.parametrize("B,N,C", [(2, 1000, 128), (1, 32, 8), (3, 70, 33)])
def test_pts_feature_one_launch(B, N, C):
    import ctypes
    from jmodt_amd import _lib as L
    g = torch.Generator().manual_seed(1)
    cls = torch.randn(B, N, 1, generator=g).to(DEV)
    xyz = (torch.rand(B, N, 3, generator=g) * 60).to(DEV)
    feats = torch.randn(B, C, N, generator=g).to(DEV)
    out = torch.empty((B, N, 2 + C), device=DEV)
    L.check(L.load().jm_pts_feature(B, N, C, ctypes.c_void_p(cls.data_ptr()), N, 1, L.dev(xyz, torch.float32, "xyz"), L.dev(feats, torch.float32, "f"),
                                    0.3, ctypes.c_void_p(out.data_ptr()), L.stream_ptr()), "pts_feature")
    assert torch.equal(out[:, :, 0], (torch.sigmoid(cls[:, :, 0]) > 0.3).float())
    depth = torch.norm(xyz.double(), p=2, dim=2) / 70.0 - 0.5
    assert float((out[:, :, 1].double() - depth).abs().max()) < 1e-6
    assert torch.equal(out[:, :, 2:], feats.transpose(1, 2))


@pytest.mark.parametrize("B,n,k1,k2,nout,act", [(8, 256, 1024, 512, 512, 1), (8, 64, 512, 1024, 256, 2), (2, 32, 12, 0, 7, 0), (3, 96, 36, 8, 4, 3),
                                                (8, 64, 1024, 1024, 1024, 1)])
def test_points_linear_vs_float64(B, n, k1, k2, nout, act):
    from jmodt_amd.ops.conv1d import points_linear
    g = torch.Generator().manual_seed(k1 + n)
    x1 = torch.randn(B, k1, n, generator=g).to(DEV)
    x2 = torch.randn(B, k2, n, generator=g).to(DEV) if k2 else None
    W = (torch.randn(nout, k1 + k2, generator=g) / np.sqrt(k1 + k2)).to(DEV)
    b = torch.randn(nout, generator=g).to(DEV)
    scale = torch.rand(B * n, 4, generator=g).to(DEV)
    x = x1.double() if x2 is None else torch.cat([x1, x2], dim=1).double()
    pre = torch.einsum("ok,bkn->bon", W.double(), x) + b.double()[None, :, None]
    want = [pre, torch.relu(pre), torch.tanh(pre), torch.sigmoid(pre)][act]
    got = points_linear(x1, W, b, act, x2=x2)
    assert float((got.double() - want).abs().max()) < 2e-5 * max(1.0, float(want.abs().max()))
    got = points_linear(x1, W, b, act, x2=x2, rowscale=scale, rowscale_stride=4, out_rows=nout + (-nout) % 4)
    want_rows = (want * scale[:, 0].double().view(B, 1, n)).transpose(1, 2).reshape(B * n, nout)
    assert float((got[:, :nout].double() - want_rows).abs().max()) < 2e-5 * max(1.0, float(want.abs().max()))


def test_decode_rpn_proposals_reads_a_channel_major_view_in_place():
    """the RPN heads' (B, 1 + C, N) output seen as a (B, N, C) view: same proposals as from the contiguous copy, bit for bit"""
    from jmodt_amd.ops.proposal import decode_rpn_proposals
    g = torch.Generator().manual_seed(3)
    B, N, C = 3, 1000, 76
    both = torch.randn(B, 1 + C, N, generator=g).to(DEV)
    xyz = (torch.rand(B, N, 3, generator=g) * 40).to(DEV)
    view = both[:, 1:].transpose(1, 2)
    assert not view.is_contiguous()
    assert torch.equal(decode_rpn_proposals(xyz, view), decode_rpn_proposals(xyz, view.contiguous()))
