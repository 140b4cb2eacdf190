"""Sharding of independent clips across the GPUs of one node.

The hot path has no exchange step: every (clip, channel) signal is independent (reference
``mdctransformer.py:292-295`` folds channels into the batch axis and every op is batched), so the
batch axis is split contiguously across ranks and outputs stay sharded.  The only collective is an
all-reduce of a handful of scalars (frame counts, checksums, timers) used as barrier and result
aggregation -- RCCL over xGMI on GPUs (``backend="nccl"``), gloo in the CPU tests.
"""

from __future__ import annotations

import os

import torch
import torch.distributed as dist


def clip_range(batches_n: int, rank: int, world_size: int):
    """Contiguous split of the clip axis: rank r gets [lo, hi); sizes differ by at most one clip."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank %d / world_size %d" % (rank, world_size))
    base, rem = divmod(int(batches_n), world_size)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init_process_group(backend=None, force=False):
    """Initialise torch.distributed from the torchrun environment.  A single process gets no group unless ``force``:
    a one-rank group sends the same barrier / all-reduce calls through the backend (RCCL on the device for "nccl") as
    an N-rank run does, which is how the multi-GPU code path is exercised on a one-GPU box."""
    rank, world, local_rank = env_rank_world()
    if world == 1 and not force:
        return rank, world, local_rank
    if not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            if torch.cuda.device_count() < world:
                raise RuntimeError("%d ranks over RCCL need %d devices, this box has %d"
                                   % (world, world, torch.cuda.device_count()))
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def reduce_scalars(values, op="sum", device=None):
    """All-reduce a short list of Python floats (float64 on the wire).  Also serves as a barrier."""
    if not (dist.is_available() and dist.is_initialized()):
        return [float(v) for v in values]
    if device is None or dist.get_backend() != "nccl":
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    ops = {"sum": dist.ReduceOp.SUM, "max": dist.ReduceOp.MAX, "min": dist.ReduceOp.MIN}
    dist.all_reduce(t, op=ops[op])
    return [float(v) for v in t.cpu().tolist()]


def barrier():
    if dist.is_available() and dist.is_initialized():
        if dist.get_backend() == "nccl":
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()
