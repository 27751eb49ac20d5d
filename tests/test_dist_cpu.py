"""CPU tier: the multi-process paths with the gloo backend, world size 2 (N > 1 coverage without
GPUs): frame sharding, bucketed gradient all-reduce, and the finetune DP step reproducing the
single-process gradient step of the whole batch (what nn.DataParallel computes in the reference,
tools/train.py:86-107)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from jmodt_amd import dist as jdist


def test_shard_frames_keeps_pairs_together():
    for frames, world in ((32, 8), (8, 8), (20, 8), (6, 4), (2, 2)):
        spans = [jdist.shard_frames(frames, world, r) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == frames
        for (b, e), (b2, _) in zip(spans, spans[1:] + [(frames, frames)]):
            assert e == b2 and b % 2 == 0 and (e - b) % 2 == 0
    with pytest.raises(ValueError):
        jdist.shard_frames(7, 2, 0)
    assert jdist.shard_frames(8, 3, 0, pair_aligned=False) == (0, 3)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make_batch(seed, frames=8, rois=12, C=32):
    g = torch.Generator().manual_seed(seed)
    feats = torch.relu(torch.randn(frames, rois, C, generator=g))
    tids = torch.randint(0, 5, (frames, rois), generator=g).float()   # 0 = background
    return feats, tids


def _make_heads(C=32):
    from jmodt_amd.ops.affinity import make_affinity_mlp
    torch.manual_seed(7)
    return make_affinity_mlp(C, (C, C)), make_affinity_mlp(C, (C, C))


def _worker(rank, world, port, tmpdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import torch.distributed as tdist
    from jmodt_amd.ops.affinity_train import finetune_step
    r, lr, w = jdist.init_from_env("gloo")
    assert (r, w) == (rank, world)
    # --- bucketed all-reduce: tiny buckets force several collectives, None grads are zero-filled
    ps = [torch.nn.Parameter(torch.full((5,), float(rank + 1))), torch.nn.Parameter(torch.ones(3, 3)),
          torch.nn.Parameter(torch.ones(7))]
    ps[0].grad = torch.full((5,), float(rank + 1)); ps[1].grad = torch.full((3, 3), 10.0 * (rank + 1))
    n = jdist.allreduce_gradients(ps, bucket_bytes=40, average=True)
    assert n == 3
    assert torch.allclose(ps[0].grad, torch.full((5,), 1.5)) and torch.allclose(ps[1].grad, torch.full((3, 3), 15.0))
    assert torch.equal(ps[2].grad, torch.zeros(7))
    # --- DP finetune step on this rank's frame-pair shard
    feats, tids = _make_batch(123)
    link, se = _make_heads()
    opt = torch.optim.SGD(list(link.parameters()) + list(se.parameters()), lr=0.1)
    b, e = jdist.shard_frames(feats.shape[0], world, rank)
    loss = finetune_step(feats[b:e], tids[b:e], link, se, opt, world=world)
    torch.save({"loss": loss, "link": link.state_dict(), "se": se.state_dict()}, os.path.join(tmpdir, f"r{rank}.pt"))
    # --- the same step in the static-shape, sync-free form (device-side counts, no .item())
    from jmodt_amd.ops.affinity_train import finetune_step_static
    link2, se2 = _make_heads()
    opt2 = torch.optim.SGD(list(link2.parameters()) + list(se2.parameters()), lr=0.1)
    loss2 = finetune_step_static(feats[b:e], tids[b:e], link2, se2, opt2, world=world)
    torch.save({"loss": float(loss2), "link": link2.state_dict(), "se": se2.state_dict()}, os.path.join(tmpdir, f"s{rank}.pt"))
    tdist.barrier()
    tdist.destroy_process_group()


def test_dp_finetune_step_matches_single_process(tmp_path):
    from jmodt_amd.ops.affinity_train import finetune_step, reid_loss, training_affinity
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    feats, tids = _make_batch(123)
    link, se = _make_heads()
    with torch.no_grad():
        ref_loss = float(reid_loss(training_affinity(feats, tids, link, se)))
    opt = torch.optim.SGD(list(link.parameters()) + list(se.parameters()), lr=0.1)
    loss1 = finetune_step(feats, tids, link, se, opt, world=1)
    assert abs(loss1 - ref_loss) < 1e-6
    r0, r1 = (torch.load(tmp_path / f"r{r}.pt") for r in range(world))
    assert abs(r0["loss"] - loss1) < 1e-5 and abs(r1["loss"] - loss1) < 1e-5
    for name, ref in (("link", link.state_dict()), ("se", se.state_dict())):
        for k, v in ref.items():
            assert torch.allclose(r0[name][k], v, atol=1e-6), (name, k)     # DP == single process
            assert torch.equal(r0[name][k], r1[name][k]), (name, k)         # replicas stay identical
    s0, s1 = (torch.load(tmp_path / f"s{r}.pt") for r in range(world))     # static-shape form: same loss, same update
    assert abs(s0["loss"] - loss1) < 1e-5 and abs(s1["loss"] - loss1) < 1e-5
    for name, ref in (("link", link.state_dict()), ("se", se.state_dict())):
        for k, v in ref.items():
            assert torch.allclose(s0[name][k], v, atol=1e-6), (name, k)
            assert torch.equal(s0[name][k], s1[name][k]), (name, k)


def _one_rank_worker(rank, port, tmpdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    import torch.distributed as tdist
    from jmodt_amd.ops.affinity_train import finetune_step_static
    tdist.init_process_group("gloo")
    assert jdist.collective_path(1) and jdist.collective_path(None) and jdist.group_world() == 1
    assert not jdist.collective_path(None, local=True)                   # the explicit opt-out inside a group
    with pytest.raises(RuntimeError, match="declared inside a process group"):
        jdist.group_world(4)                                             # a declaration that differs from the group's size
    ps = [torch.nn.Parameter(torch.ones(5)), torch.nn.Parameter(torch.ones(3, 3))]
    ps[0].grad = torch.arange(5.0)
    assert jdist.allreduce_gradients(ps, bucket_bytes=16) == 2           # ISSUED on the one-rank group (not skipped)
    assert torch.equal(ps[0].grad, torch.arange(5.0)) and torch.equal(ps[1].grad, torch.zeros(3, 3))
    feats, tids = _make_batch(123)
    link, se = _make_heads()
    opt = torch.optim.SGD(list(link.parameters()) + list(se.parameters()), lr=0.1)
    loss = finetune_step_static(feats, tids, link, se, opt, world=1)
    torch.save({"loss": float(loss), "link": link.state_dict(), "se": se.state_dict()}, os.path.join(tmpdir, "one.pt"))
    tdist.destroy_process_group()


def test_one_rank_process_group_takes_the_collective_path_and_changes_nothing(tmp_path):
    """the data-parallel step issues its collectives whenever a process group exists, world size 1 included (what
    `bench.py --launch` runs on a one-GPU box): counts, gradients and loss go through all_reduce, and the parameters after
    the step equal the step without any process group bit for bit; a declared world > 1 without a group is an error"""
    from jmodt_amd.ops.affinity_train import finetune_step_static
    mp.spawn(_one_rank_worker, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    assert not jdist.collective_path(1)
    with pytest.raises(RuntimeError):
        jdist.collective_path(2)
    feats, tids = _make_batch(123)
    link, se = _make_heads()
    opt = torch.optim.SGD(list(link.parameters()) + list(se.parameters()), lr=0.1)
    loss = float(finetune_step_static(feats, tids, link, se, opt, world=1))
    one = torch.load(tmp_path / "one.pt")
    assert one["loss"] == loss
    for name, ref in (("link", link.state_dict()), ("se", se.state_dict())):
        for k, v in ref.items():
            assert torch.equal(one[name][k], v), (name, k)


def _joint_bucket_worker(rank, world, port, tmpdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as tdist
    from jmodt_amd.detector import DetectAffinityEngine
    tdist.init_process_group("gloo")
    torch.manual_seed(0)
    eng = DetectAffinityEngine()                                    # the reference widths: 16 732 011 parameters = 66.9 MB
    params = list(eng.parameters())
    for k, p in enumerate(params):                                  # a rank-dependent "gradient" with structure per tensor
        p.grad = torch.full_like(p, float(rank + 1)) * (1.0 + (k % 7))
    n = jdist.allreduce_gradients(params, world=world, bucket_bytes=64 << 20, average=False)
    ok = all(torch.equal(p.grad, torch.full_like(p, 3.0) * (1.0 + (k % 7))) for k, p in enumerate(params))
    n4 = jdist.allreduce_gradients(params, world=world, bucket_bytes=16 << 20, average=True)      # smaller buckets, mean
    ok = ok and n4 >= 4 and all(torch.equal(p.grad, torch.full_like(p, 3.0) * (1.0 + (k % 7))) for k, p in enumerate(params))
    torch.save({"collectives": n, "ok": ok, "bytes": sum(p.numel() for p in params) * 4}, os.path.join(tmpdir, f"j{rank}.pt"))
    tdist.barrier()
    tdist.destroy_process_group()


def test_joint_mode_gradient_exchange_of_all_parameters(tmp_path):
    """BASELINE configs[3]'s second message size: the gradient of ALL parameters of the detector + affinity
    heads, 66.9 MB fp32, fits ONE 64 MiB bucket (one collective; a 16 MiB bucket size makes it four) and comes back as the SUM over ranks (world size 2, gloo)"""
    world = 2
    mp.spawn(_joint_bucket_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        res = torch.load(tmp_path / f"j{r}.pt")
        assert res["ok"] and res["collectives"] == 1 and res["bytes"] == 66_928_044, res


def test_static_training_affinity_equals_the_looped_form():
    """training_affinity_static (masks instead of torch.unique / a Python loop over frame pairs) gives the reference
    form's loss and gradients exactly (float64), incl. a pair without foreground on one side and duplicate track ids"""
    from jmodt_amd.ops.affinity import make_affinity_mlp
    from jmodt_amd.ops.affinity_train import reid_loss, reid_loss_static, training_affinity, training_affinity_static
    g = torch.Generator().manual_seed(5)
    frames, R, C = 8, 16, 32
    feats = torch.relu(torch.randn(frames, R, C, generator=g)).double()
    tids = torch.randint(0, 5, (frames, R), generator=g).double()
    tids[2] = 0                      # pair 1: no foreground in prev -> skipped
    tids[5] = 3                      # pair 2: every next RoI on one track
    torch.manual_seed(1)
    link, se = make_affinity_mlp(C, (C, C)).double(), make_affinity_mlp(C, (C, C)).double()
    params = list(link.parameters()) + list(se.parameters())
    a = training_affinity(feats, tids, link, se)
    la = reid_loss(a)
    la.backward()
    ga = [p.grad.clone() for p in params]
    for p in params:
        p.grad = None
    b = training_affinity_static(feats, tids, link, se)
    lb, counts = reid_loss_static(b)
    lb.backward()
    assert counts.tolist() == [a["gt_links"].numel(), a["gt_starts"].numel(), a["gt_ends"].numel()]
    assert abs(la.item() - lb.item()) < 1e-12
    assert max((x - p.grad).abs().max().item() for x, p in zip(ga, params)) < 1e-12
    # the valid entries ARE the reference's matrices: same multiset of link scores / labels
    assert torch.allclose(torch.sort(b["link"][b["valid"]])[0], torch.sort(a["rcnn_link"].view(-1))[0], atol=1e-12)
    assert b["gt_links"][b["valid"]].sum().item() == a["gt_links"].sum().item()


def test_training_affinity_matches_reference_shapes_and_labels():
    from jmodt_amd.ops.affinity_train import get_unique_tid_feature, training_affinity
    feats = torch.arange(24, dtype=torch.float32).view(2, 3, 4)
    tids = torch.tensor([[3., 0., 3.], [5., 3., 0.]])
    u, f = get_unique_tid_feature(tids[0][tids[0] > 0], feats[0][tids[0] > 0])
    assert u.tolist() == [3.0] and torch.equal(f[0], (feats[0, 0] + feats[0, 2]) / 2)
    link, se = _make_heads(4)
    out = training_affinity(feats, tids, link, se)
    assert out["rcnn_link"].shape == (2, 1) and out["gt_links"].tolist() == [1.0, 0.0]   # prev {3} x next {3, 5}
    assert out["gt_starts"].tolist() == [0.0, 1.0] and out["gt_ends"].tolist() == [0.0]
    assert out["rcnn_start"].shape == (2, 1) and out["rcnn_end"].shape == (1, 1)
    # 1 x 2 score matrix: the row softmax sums to 1, each single-entry column softmax is 1 -> (1 + 2) / 2
    assert abs(float(out["rcnn_link"].detach().sum()) - 1.5) < 1e-5
