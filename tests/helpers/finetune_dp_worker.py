"""helper process of tests/test_gpu_surface.py::test_finetune_step_hip_data_parallel_equals_single_process.

    RANK=r WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=p python tests/helpers/finetune_dp_worker.py <out.pt>     (one per rank)
    python tests/helpers/finetune_dp_worker.py <out.pt> --single

one finetune step of the link / start-end heads on the HAND-WRITTEN training kernels (csrc/affinity_train.hip) on this rank's
pair-aligned frame shard: the three loss-mean element counts, the flat gradient bucket and the loss all-reduced over gloo on device
tensors (both ranks share cuda:0 on the one-GPU box), SGD.  Writes the updated parameters and the whole-batch loss."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    out, single = sys.argv[1], "--single" in sys.argv
    from jmodt_amd import dist as jdist
    from jmodt_amd.ops.affinity import make_affinity_mlp
    from jmodt_amd.ops.affinity_train import finetune_step_static
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    rank, world = (0, 1) if single else (int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]))
    if not single:
        import torch.distributed as tdist
        tdist.init_process_group("gloo")
    torch.manual_seed(0)
    link, se = make_affinity_mlp(128, (96, 64)).to(dev).train(), make_affinity_mlp(128, (96, 64)).to(dev).train()
    params = list(link.parameters()) + list(se.parameters())
    opt = torch.optim.SGD(params, lr=0.05)
    g = torch.Generator().manual_seed(1)
    frames, R = 8, 64
    feats = torch.relu(torch.randn(frames, R, 128, generator=g))
    tids = torch.randint(0, 9, (frames, R), generator=g).float()
    tids[2] = 0                                  # a pair without foreground on one side (skipped, rcnn.py:230) — on rank 0's shard
    b, e = jdist.shard_frames(frames, world, rank)
    loss = finetune_step_static(feats[b:e].to(dev), tids[b:e].to(dev), link, se, opt, world=world)
    torch.cuda.synchronize()
    torch.save({"params": [p.detach().cpu() for p in params], "loss": float(loss), "frames": (b, e)}, out)
    if not single:
        import torch.distributed as tdist
        tdist.barrier()
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()
