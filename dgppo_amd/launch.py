"""Self-launcher for the data-parallel drivers (`bench.py --gpus N`, `train.py --gpus N` without a torchrun environment):
one child process per rank, started BEFORE the parent makes any GPU call (the parent never initialises HIP and nothing
is exec'ed after GPU init), supervised until all of them have exited.

What the supervision guarantees (the first multi-GPU run must end as a diagnosable failure, never as
"killed at the limit, nothing written"):
  * every rank's stdout / stderr go to files (`rank{r}.out`, `rank{r}.err`) in `log_dir`; rank 0's stderr is relayed to the
    parent's stderr while it runs (progress lines), its stdout to the parent's stdout at the end;
  * all children are polled; when ANY rank exits non-zero the others are terminated (SIGTERM, SIGKILL after a grace
    period) — a rank blocked in a collective whose peer died would otherwise sit there until an outer time limit — and the
    parent returns non-zero after printing the failing rank and the tail of its stderr;
  * a stall deadline: when no rank has written anything for `stall_seconds`, the job is taken to be hung in a collective,
    every rank is stopped and the tails of all stderr files are printed;
  * rendezvous of the gloo control plane goes through a file store (`DGPPO_RDZV_FILE`), not a port picked by bind/close.
"""
from __future__ import annotations

import os
import signal
import subprocess
import sys
import tempfile
import time
from typing import Dict, List, Optional


def _tail(path: str, n: int = 30) -> str:
    try:
        with open(path, "rb") as f:
            f.seek(0, os.SEEK_END)
            size = f.tell()
            f.seek(max(0, size - 16384))
            lines = f.read().decode("utf-8", "replace").splitlines()
        return "\n".join("    | " + ln for ln in lines[-n:])
    except OSError as ex:
        return f"    | <cannot read {path}: {ex}>"


def _stop_all(procs: List[subprocess.Popen], grace_s: float = 5.0) -> None:
    """terminate exactly the children this launcher started (by PID, never by pattern)"""
    for p in procs:
        if p.poll() is None:
            try:
                p.send_signal(signal.SIGTERM)
            except OSError:
                pass
    t_end = time.time() + grace_s
    for p in procs:
        while p.poll() is None and time.time() < t_end:
            time.sleep(0.05)
        if p.poll() is None:
            try:
                p.kill()
            except OSError:
                pass
            p.wait()


def default_log_dir(root: str) -> str:
    d = os.path.join(root, "gpurun_out")
    return d if os.path.isdir(d) else tempfile.mkdtemp(prefix="dgppo_ranks_")


def spawn_ranks(script: str, argv: List[str], world: int, log_dir: str, stall_seconds: float = 420.0,
                poll_s: float = 0.2, extra_env: Optional[Dict[str, str]] = None, out=sys.stdout, err=sys.stderr) -> int:
    """run `python script argv...` as `world` ranks (RANK / LOCAL_RANK / WORLD_SIZE in the environment) and supervise them.
    -> 0 when every rank exited 0, else 1."""
    os.makedirs(log_dir, exist_ok=True)
    rdzv = os.path.join(tempfile.mkdtemp(prefix="dgppo_rdzv_"), "store")
    base = dict(os.environ)
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
    base.setdefault("NCCL_DEBUG", "WARN")                    # RCCL's own diagnostics land in rank{r}.err
    base.update(extra_env or {})
    procs, files = [], []
    for r in range(world):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=base.get("MASTER_PORT", "29500"), DGPPO_RDZV_FILE=rdzv)
        fo = open(os.path.join(log_dir, f"rank{r}.out"), "wb")
        fe = open(os.path.join(log_dir, f"rank{r}.err"), "wb")
        files.append((fo, fe))
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env, stdout=fo, stderr=fe))
    err_paths = [os.path.join(log_dir, f"rank{r}.err") for r in range(world)]
    out_paths = [os.path.join(log_dir, f"rank{r}.out") for r in range(world)]
    relayed = 0
    sizes = [-1] * (2 * world)
    last_activity = time.time()
    failure = None

    def relay_rank0():
        nonlocal relayed
        try:
            with open(err_paths[0], "rb") as f:
                f.seek(relayed)
                chunk = f.read()
        except OSError:
            return
        if chunk:
            relayed += len(chunk)
            err.write(chunk.decode("utf-8", "replace"))
            err.flush()

    try:
        while True:
            rcs = [p.poll() for p in procs]
            relay_rank0()
            now = time.time()
            cur = [os.path.getsize(pth) if os.path.exists(pth) else 0 for pth in err_paths + out_paths]
            if cur != sizes:
                sizes, last_activity = cur, now
            bad = [(r, rc) for r, rc in enumerate(rcs) if rc not in (None, 0)]
            if bad:
                failure = f"rank(s) exited non-zero (rank, exit code): {bad}"
                break
            if all(rc == 0 for rc in rcs):
                break
            if now - last_activity > stall_seconds:
                alive = [r for r, rc in enumerate(rcs) if rc is None]
                failure = (f"no rank wrote anything for {stall_seconds:.0f} s (still running: {alive}) — taken to be hung in a "
                           f"collective or rendezvous")
                break
            time.sleep(poll_s)
    finally:
        _stop_all(procs)
        for fo, fe in files:
            fo.close(); fe.close()
    relay_rank0()
    try:
        with open(out_paths[0], "rb") as f:
            out.write(f.read().decode("utf-8", "replace"))
            out.flush()
    except OSError:
        pass
    if failure is None:
        return 0
    print(f"[launch] FAILED: {failure}; all ranks stopped.  Per-rank logs: {log_dir}/rank*.err", file=err)
    rcs = [p.returncode for p in procs]
    first_bad = [r for r, rc in enumerate(rcs) if rc not in (0, None, -signal.SIGTERM, -signal.SIGKILL)]
    for r in (first_bad or list(range(world))):
        print(f"[launch] rank {r} (exit code {rcs[r]}), last stderr lines:\n{_tail(err_paths[r])}", file=err)
    err.flush()
    return 1
