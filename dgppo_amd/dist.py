"""Data-parallel plumbing: one process per GPU, `torch.distributed` over RCCL (backend "nccl" on ROCm; "gloo" in the CPU
tests).  The hot path shards naturally: every rank owns its own envs, rollouts, value pre-passes, GAE and
advantages with NO communication; the only exchange is one all-reduce(sum) of each network's flat fp32 gradient buffer per
minibatch step, after which every rank applies the identical clip + Adam update (replicas stay bit-identical)."""
from __future__ import annotations

import os
from typing import Callable, Optional

import numpy as np
import torch


def env_info():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend: Optional[str] = None, device: Optional[torch.device] = None):
    """idempotent init from the torchrun environment; returns (rank, world)."""
    import torch.distributed as dist
    rank, local_rank, world = env_info()
    if world > 1 and not dist.is_initialized():
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, **kw)
    return rank, world


def make_allreduce(world: int) -> Optional[Callable[[torch.Tensor], None]]:
    """gradient hook for Engine: sum over ranks, then scale by 1/world (losses are means over equal-sized shards, so the
    mean of the shard gradients is the gradient of the global mean)."""
    if world <= 1:
        return None
    import torch.distributed as dist

    def allreduce(flat: torch.Tensor):
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.mul_(1.0 / world)
    return allreduce


def shard_seeds(rank: int, B_local: int, iteration: int, run_seed: int = 0) -> np.ndarray:
    """scene seeds of this rank's envs: a function of the GLOBAL env index, so any (rank, world) split of the same global
    batch sees the same scenes."""
    gidx = np.arange(B_local, dtype=np.uint64) + np.uint64(rank * B_local + 1)
    s = gidx * np.uint64(0x9E3779B97F4A7C15)
    return (s ^ np.uint64(run_seed) ^ np.uint64((iteration * 7919 + 1) & 0xFFFFFFFF)).view(np.int64)


def max_over_ranks(x: float, world: int, device) -> float:
    if world <= 1:
        return x
    import torch.distributed as dist
    t = torch.tensor([x], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
