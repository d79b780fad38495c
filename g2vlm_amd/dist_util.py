"""Multi-GPU plumbing for the bench / serving harness: one process per GPU, torch.distributed
(backend "nccl" = RCCL on ROCm; "gloo" for the CPU rehearsal in tests).

Round-1 scope: scenes are independent, so N ranks run N replicas of the
single-GPU path with NO data-path collective; the only exchanges are the timing barrier and the
max-over-ranks reduction of the wall time.  The view-sharded 32-view path (one KV
all-gather per MoT layer) partitions views with `shard_views`.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend="nccl", device=None):
    world, rank, local = env_world()
    if world > 1 and not dist.is_initialized():
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, **kw)
    return world, rank, local


def barrier(sync_cuda=True):
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
    if sync_cuda and torch.cuda.is_available():
        torch.cuda.synchronize()


def max_over_ranks(seconds, device="cpu"):
    """Wall time of the slowest rank (the whole-job time of a weak-scaling replica run)."""
    if not (dist.is_available() and dist.is_initialized()):
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])


def aggregate_throughput(units_per_rank, seconds, device="cpu"):
    """value = units all ranks processed / max-over-ranks time."""
    world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    return world * units_per_rank / max_over_ranks(seconds, device)


def shard_views(n_views, world, rank):
    """Contiguous view range [lo, hi) owned by `rank` (rank r owns views [4r, 4r+4) at 32 views / 8 GPUs)."""
    base, rem = divmod(n_views, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
