"""`bench.py --gpus N` on the GPU box: the self-launched, supervised multi-rank path (two ranks sharing the one GPU over the
gloo data plane — RCCL needs one device per rank and multi-GPU boxes are the driver's), small sizes."""
import json
import os
import subprocess
import sys
import time

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, *flags, timeout=600):
    env = dict(os.environ, DGPPO_DIST_BACKEND="gloo", **extra_env)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "DGPPO_RDZV_FILE"):
        env.pop(k, None)
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--n-env", "256", "--batch-size", "4096",
                        "--steps", "1", "--warmup", "1", "--no-cpu-baseline", *flags], capture_output=True, text=True, env=env,
                       timeout=timeout, cwd=ROOT)
    return r, time.time() - t0


def test_two_supervised_ranks_print_one_json_line(cuda):
    r, _ = _run({})
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["config"]["global_envs"] == 512
    assert out["value"] == pytest.approx(2 * 256 * 128 / (out["ms_per_step"] * 1e-3), rel=1e-6)
    assert out["multi_gpu"]["allreduce_selfcheck"] == "passed"
    assert "all-reduce self-check passed" in r.stderr and "multi-GPU diagnostics" in r.stderr


def test_a_rank_that_dies_takes_the_job_down_within_seconds(cuda):
    """DGPPO_BENCH_FAULT_RANK=1: rank 1 exits (code 3) once the communicator is up; rank 0 goes on into its first all-reduce
    and would wait there for its peer.  The supervising parent must stop it and return non-zero with the reason."""
    r, dt = _run({"DGPPO_BENCH_FAULT_RANK": "1"})
    assert r.returncode != 0
    assert "[launch] FAILED" in r.stderr and "(1, 3)" in r.stderr, r.stderr[-3000:]
    assert dt < 240, f"took {dt:.0f} s"            # dominated by two `import torch` + engine construction, not by a timeout
