"""helper process of tests/test_gpu_train_joint.py::test_joint_step_data_parallel_equals_single_process.

    RANK=r WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=p python tests/helpers/joint_dp_worker.py <out.pt>      (one per rank)
    python tests/helpers/joint_dp_worker.py <out.pt> --single                                                   (the whole batch)
    ... --rcnn: the RPN-fixed step instead (train_joint.rcnn_forward_backward: frozen fused RPN, RCNN + re-id heads on the row kernels;
    only the RCNN's gradients are exchanged)

forward / thin loss / backward of the joint-mode step (jmodt_amd/train_joint.py) on this rank's pair-aligned frame shard, the
re-id element counts and the gradients of ALL parameters all-reduced (SUM) — over gloo, whose collectives take device tensors:
both ranks share cuda:0 on the one-GPU box.  Writes the reduced gradients."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    out, single, rcnn = sys.argv[1], "--single" in sys.argv, "--rcnn" in sys.argv
    from jmodt_amd import dist as jdist, synth
    from jmodt_amd.detector import DetectorConfig
    from jmodt_amd.ops.affinity_train import AffinityTrainState
    from jmodt_amd.train_joint import joint_forward, prepare_rcnn, rcnn_forward_backward, rcnn_parameters, thin_loss
    from tests.test_gpu_detector import make_engine
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    rank, world = (0, 1) if single else (int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]))
    if not single:
        import torch.distributed as tdist
        tdist.init_process_group("gloo")
    eng = make_engine(seed=3, cfg=DetectorConfig.tiny()).to(dev).eval()       # eval-mode BatchNorm: no per-shard batch statistics
    for p in eng.parameters():
        p.requires_grad_(True)
    frames = 4
    xyz, img, xy = synth.frames(frames, 2048, 77, H=96, W=320, native=(94, 310))
    R = min(64, eng.cfg.rpn_post_nms_top_n)
    tids = torch.randint(0, 6, (frames, R), generator=torch.Generator().manual_seed(4)).float()
    b, e = jdist.shard_frames(frames, world, rank)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a[b:e])).to(dev)     # noqa: E731
    tids = tids[b:e].to(dev)
    if rcnn:
        prepare_rcnn(eng)
        for m in eng.modules():                 # (Dropout draws per process: off, so that the shards' gradients are comparable)
            if isinstance(m, torch.nn.Dropout):
                m.eval()
        loss, _ = rcnn_forward_backward(eng, T(xyz), T(img), T(xy), tids, None if single else world, single, R)
        n = jdist.allreduce_gradients(rcnn_parameters(eng), world=world, bucket_bytes=1 << 20, average=False)
        torch.save({"grads": {k: p.grad.detach().cpu() for k, p in eng.named_parameters() if p.grad is not None},
                    "collectives": n, "loss": float(loss), "frames": (b, e)}, out)
        if not single:
            import torch.distributed as tdist
            tdist.barrier()
            tdist.destroy_process_group()
        return
    with torch.enable_grad():
        o = joint_forward(eng, T(xyz), T(img), T(xy), rois_per_frame=R)
        counts = None
        if jdist.collective_path(world):
            import torch.distributed as tdist
            counts = AffinityTrainState(o["rcnn_feat"].detach().view(e - b, -1, o["rcnn_feat"].shape[-1]), tids).counts.clone()
            tdist.all_reduce(counts, op=tdist.ReduceOp.SUM)
        loss = thin_loss(eng, o, tids, counts)
    loss.backward()
    params = [p for p in eng.parameters() if p.requires_grad]
    n = jdist.allreduce_gradients(params, world=world, bucket_bytes=1 << 20, average=False)
    torch.save({"grads": {k: p.grad.detach().cpu() for k, p in eng.named_parameters() if p.grad is not None},
                "collectives": n, "loss": float(loss.detach()), "frames": (b, e)}, out)
    if not single:
        import torch.distributed as tdist
        tdist.barrier()
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()
