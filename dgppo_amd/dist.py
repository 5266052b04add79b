"""Data-parallel plumbing: one process per GPU.  The hot path shards naturally: every rank owns its own envs,
rollouts, value pre-passes, GAE and advantages with NO communication; the only exchange is ONE all-reduce(sum) per
minibatch of the engine's flat `[g_policy | g_Vl | g_Vh | loss sums]` fp32 buffer, after which every rank applies the
identical NaN-check -> norm -> clip -> Adam with `grad_scale = 1/world` (replicas stay bit-identical).

Two planes:
  * data plane   — `RcclComm`: the C-ABI entry points `dgppo_comm_{unique_id,init,allreduce_sum_f32,destroy}` of
                   libdgppo_hip.so (RCCL over xGMI, enqueued on the caller's HIP stream; include/dgppo_hip.h §C1);
  * control plane — a `torch.distributed` gloo group on the host: ships the 128-byte rendezvous id, barriers, and the
                   max-over-ranks of the benchmark clock.  (backend "gloo" also serves as a data plane for rehearsals of
                   several ranks on ONE GPU, where RCCL refuses duplicate devices, and for the CPU tests.)
"""
from __future__ import annotations

import ctypes as C
import datetime
import os
from typing import Callable, Optional

import numpy as np
import torch

ID_BYTES = 128


def env_info():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_control_plane(timeout_s: int = 600):
    """idempotent gloo group from the torchrun-style environment (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT);
    returns (rank, world)."""
    import torch.distributed as dist
    rank, _, world = env_info()
    if world > 1 and not dist.is_initialized():
        # DGPPO_RDZV_FILE (set by the self-launching drivers, `bench.py --gpus N` / `train.py --gpus N`): rendezvous through a
        # file store instead of a TCP port picked by bind/close, which another process could take in between
        rdzv = os.environ.get("DGPPO_RDZV_FILE")
        kw = dict(init_method=f"file://{rdzv}", rank=rank, world_size=world) if rdzv else {}
        # gloo reports its mesh ("[Gloo] Rank 0 is connected to ...") on fd 1; the drivers promise ONE JSON line on stdout, so
        # fd 1 points at stderr while the group forms
        import sys
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=timeout_s), **kw)
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
    return rank, world


def rccl_version() -> Optional[int]:
    """ncclGetVersion of the RCCL the C ABI resolves (e.g. 22204), or None when no RCCL library can be loaded"""
    from . import _native as N
    v = C.c_int32(0)
    return int(v.value) if N.lib().dgppo_comm_version(C.byref(v)) == 0 else None


def selfcheck_allreduce(allreduce: Optional[Callable], rank: int, world: int, device, count: int = 4099) -> None:
    """one tiny all-reduce with a known answer before anything is timed or trained on: every rank contributes rank + 1 in
    every entry, so every entry must come back as world (world + 1) / 2.  Raises on any rank that sees something else."""
    if allreduce is None or world <= 1:
        return
    x = torch.full((count,), float(rank + 1), device=device)
    allreduce(x)
    got = x.cpu()
    want = world * (world + 1) / 2
    if not bool((got == want).all()):
        bad = int((got != want).sum())
        raise RuntimeError(f"all-reduce self-check failed on rank {rank}/{world}: expected {want} in all {count} entries, "
                           f"{bad} differ (first values {got[:4].tolist()})")


class RcclComm:
    """RCCL communicator behind the C ABI.  Construct on every rank AFTER `torch.cuda.set_device` (the communicator binds
    to the current HIP device) and after `init_control_plane()`."""

    def __init__(self, rank: int, world: int):
        import torch.distributed as dist
        from . import _native as N
        self._N, self.rank, self.world = N, rank, world
        lib = N.lib()
        buf = (C.c_uint8 * ID_BYTES)()
        if rank == 0:
            N.check(lib.dgppo_comm_unique_id(buf), "dgppo_comm_unique_id")
        if world > 1:
            idt = torch.tensor(list(buf), dtype=torch.uint8)
            dist.broadcast(idt, src=0)                               # host tensor over the gloo control plane
            for i, b in enumerate(idt.tolist()):
                buf[i] = b
        self._handle = C.c_void_p()
        N.check(lib.dgppo_comm_init(buf, C.c_int32(rank), C.c_int32(world), C.byref(self._handle)), "dgppo_comm_init")

    def allreduce_sum(self, flat: torch.Tensor) -> None:
        """in-place sum over the ranks, enqueued on torch's current stream"""
        N = self._N
        rc = N.lib().dgppo_comm_allreduce_sum_f32(self._handle, N.ptr(flat), C.c_int64(flat.numel()), N.stream_ptr())
        N.check(rc, "dgppo_comm_allreduce_sum_f32")

    def destroy(self) -> None:
        if self._handle:
            self._N.check(self._N.lib().dgppo_comm_destroy(self._handle), "dgppo_comm_destroy")
            self._handle = C.c_void_p()


def make_allreduce(world: int, backend: str = "rccl"):
    """-> (allreduce(flat) or None, closer()).  backend "rccl": the C-ABI communicator; "gloo": torch.distributed on the
    control-plane group (rehearsals on one GPU / CPU tests).  The callable SUMS in place; the engine divides by `world`
    inside the optimiser kernel (`grad_scale`), so there is no scaling pass here."""
    if world <= 1:
        return None, (lambda: None)
    import torch.distributed as dist
    rank, _ = init_control_plane()
    if backend == "rccl":
        comm = RcclComm(rank, world)
        return comm.allreduce_sum, comm.destroy
    if backend != "gloo":
        raise ValueError(f"unknown data-plane backend '{backend}' (rccl | gloo)")

    def allreduce(flat: torch.Tensor):
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return allreduce, (lambda: None)


def shard_seeds(rank: int, B_local: int, iteration: int, run_seed: int = 0) -> np.ndarray:
    """scene seeds of this rank's envs: a function of the GLOBAL env index, so any (rank, world) split of the same global
    batch sees the same scenes."""
    gidx = np.arange(B_local, dtype=np.uint64) + np.uint64(rank * B_local + 1)
    s = gidx * np.uint64(0x9E3779B97F4A7C15)
    return (s ^ np.uint64(run_seed) ^ np.uint64((iteration * 7919 + 1) & 0xFFFFFFFF)).view(np.int64)


def barrier(world: int) -> None:
    if world > 1:
        import torch.distributed as dist
        dist.barrier()


def max_over_ranks(x: float, world: int) -> float:
    if world <= 1:
        return x
    import torch.distributed as dist
    t = torch.tensor([x], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def host_allreduce(x: np.ndarray, op: str, world: int) -> np.ndarray:
    """"sum" / "max" of a small float64 host array over the control plane (identity for one rank)"""
    if world <= 1:
        return x
    import torch.distributed as dist
    t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64))
    dist.all_reduce(t, op={"sum": dist.ReduceOp.SUM, "max": dist.ReduceOp.MAX}[op])
    return t.numpy()


def shutdown(world: int) -> None:
    if world > 1:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()
