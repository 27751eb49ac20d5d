"""helper process of tests/test_gpu_surface.py::test_gradient_allreduce_runs_on_rccl_with_one_rank.

    python tests/helpers/rccl_one_rank_step.py <out.pt> [--group]

runs THREE finetune steps of the full-size link / start-end heads (2 x 525 825 parameters = 4 206 600 gradient bytes) on cuda:0
through the hand-written training kernels; with --group a one-rank RCCL process group exists and the step therefore issues its
collectives (counts, the flat gradient bucket, the loss) through RCCL.  Writes the parameters, the losses and what the profiler
recorded for the gradient all-reduce."""
import os
import socket
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    out, group = sys.argv[1], "--group" in sys.argv
    from jmodt_amd.ops import affinity_train
    from jmodt_amd.ops.affinity import make_affinity_mlp
    from jmodt_amd.profile import prof
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    if group:
        import torch.distributed as dist
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
            sk.bind(("127.0.0.1", 0))
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(sk.getsockname()[1]), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        dist.init_process_group("nccl", device_id=dev)
    torch.manual_seed(0)
    link, se = make_affinity_mlp().to(dev).train(), make_affinity_mlp().to(dev).train()
    params = list(link.parameters()) + list(se.parameters())
    opt = torch.optim.Adam(params, lr=1e-3)
    g = torch.Generator().manual_seed(1)
    feats = torch.relu(torch.randn(4, 64, 512, generator=g)).to(dev)
    tids = torch.randint(0, 13, (4, 64), generator=g).float().to(dev)
    prof.reset()
    prof.enabled = True
    losses = [affinity_train.finetune_step_static(feats, tids, link, se, opt, world=1) for _ in range(3)]
    torch.cuda.synchronize()
    prof.enabled = False
    rows = {r["kernel"]: r for r in prof.summary(3, 8000.0, 157.3)}
    ar = rows.get("grad_allreduce(RCCL)", {})
    torch.save({"params": [p.detach().cpu() for p in params], "losses": [float(x) for x in losses],
                "issued": affinity_train.LAST_GRAD_COLLECTIVES, "ms_per_step": ar.get("ms_per_step"),
                "bytes_per_step": ar.get("algo_bytes_per_step")}, out)
    if group:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
