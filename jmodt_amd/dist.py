"""Data-parallel plumbing: one process per GPU, RCCL over xGMI through torch.distributed.

The reference's only multi-GPU mechanism is single-process `nn.DataParallel`
(tools/train.py:86-87): scatter the batch on dim 0, replicate weights every step, gather outputs,
reduce gradients on device 0.  The MI355X-native equivalent is one process per GPU:

  * inference / every `jmodt.ops` op: frames are independent -> each rank owns a contiguous block
    of frames, NO data-path collective;
  * training: the only exchange is the gradient all-reduce (4.2 MB in finetune mode, 66.9 MB
    joint).  Gradients are flattened into a few large buckets so each collective
    is big enough to stream over all seven xGMI links of a rank; a 4.2 MB model is ONE bucket.

Backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests (world size 2).
"""
import os
from typing import Iterable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """torchrun-style rendezvous (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT).
    Returns (rank, local_rank, world_size); a no-op single-process world when WORLD_SIZE is unset."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    return rank, local_rank, world


def shard_frames(num_frames: int, world: int, rank: int, pair_aligned: bool = True) -> Tuple[int, int]:
    """[begin, end) of this rank's contiguous frame block.  With pair_aligned the split happens in
    units of (prev, next) frame pairs: the collate interleaves pairs on dim 0
    (kitti_dataset.py:440-445) and rcnn.py:212-217 de-interleaves with stride 2, so a pair must
    never straddle two ranks and every shard holds an even number of frames."""
    unit = 2 if pair_aligned else 1
    if num_frames % unit:
        raise ValueError(f"{num_frames} frames cannot be split into (prev, next) pairs")
    units = num_frames // unit
    base, extra = divmod(units, world)
    begin = rank * base + min(rank, extra)
    count = base + (1 if rank < extra else 0)
    return begin * unit, (begin + count) * unit


def group_world(world: Optional[int] = None) -> int:
    """the number of ranks a data-parallel step runs over: the process group's size whenever one exists (`world` = None, the
    default of every step function, or an int that must then EQUAL it: a caller that declares 1 inside an 8-rank group gets an
    error, not an 8-rank sum divided by 1), else the declared `world` (None = 1)"""
    if dist.is_initialized():
        size = dist.get_world_size()
        if world is not None and int(world) != size:
            raise RuntimeError(f"world={world} declared inside a process group of {size} ranks: pass world=None (= the group's size), "
                               f"or local=True for a step without collectives")
        return size
    return 1 if world is None else int(world)


def collective_path(world: Optional[int] = None, local: bool = False) -> bool:
    """do the data-parallel steps issue their collectives?  Yes whenever a process group exists (world size 1 included) unless the
    caller opts out with local=True; a declared world that differs from the group's size, or a declared world > 1 without a
    group, is an error, not a silently different step"""
    if local:
        return False
    world = group_world(world)
    if dist.is_initialized():
        return True
    if world > 1:
        raise RuntimeError(f"a data-parallel step over {world} ranks needs an initialised torch.distributed process group")
    return False


def _buckets(params: Sequence[torch.nn.Parameter], bucket_bytes: int) -> List[List[torch.nn.Parameter]]:
    out, cur, size = [], [], 0
    for p in params:
        nbytes = p.numel() * p.element_size()
        if cur and size + nbytes > bucket_bytes:
            out.append(cur)
            cur, size = [], 0
        cur.append(p)
        size += nbytes
    if cur:
        out.append(cur)
    return out


@torch.no_grad()
def allreduce_gradients(params: Iterable[torch.nn.Parameter], world: Optional[int] = None,
                        bucket_bytes: int = 64 << 20, average: bool = True, local: bool = False) -> int:
    """Sum (or average) .grad over all ranks with a few large flat collectives.

    bucket_bytes defaults to 64 MiB: xGMI is point-to-point (7 links x ~153 GB/s per GPU), so a
    collective only reaches link bandwidth when each of the 7 peers receives megabytes; the whole
    finetune gradient (4.2 MB) or the joint model (66.9 MB) fits one or two buckets.
    Parameters without a gradient contribute zeros (all ranks must issue identical collectives).
    Returns the number of collectives issued.

    The collectives are issued whenever a process group EXISTS, a one-rank group included (RCCL / gloo accept it; the sum over
    one rank is the identity, bit for bit): flatten -> all_reduce -> unflatten is the same code at every world size, so what a
    one-GPU box executes under `bench.py --launch` is what an 8-GPU node executes.  Without a process group (a plain
    single-process run), or with local=True, nothing is issued.  `world` is only a DECLARATION that is checked against the
    group (group_world): the mean divides by the group's size, never by a number the caller passed."""
    plist = [p for p in params if p.requires_grad]
    if not collective_path(world, local) or not plist:
        return 0
    size = dist.get_world_size()
    n = 0
    for bucket in _buckets(plist, bucket_bytes):
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        if average:
            flat.div_(size)
        off = 0
        for p in bucket:
            k = p.numel()
            if p.is_contiguous():
                p.grad = flat[off:off + k].view_as(p)       # the reduced bucket IS the gradient storage: no copy back (252 launches
            else:                                           # per joint-mode step); other memory formats keep their own layout
                if p.grad is None:
                    p.grad = torch.empty_like(p)
                p.grad.copy_(flat[off:off + k].view_as(p))
            off += k
        n += 1
    return n


def max_over_ranks(value: float, device: torch.device) -> float:
    """wall-clock bracketing helper for bench.py: max of a scalar over ranks"""
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
