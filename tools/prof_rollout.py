"""Developer tool: wall time of Engine.rollout (collect) so that it can be compared with the summed kernel time of a
`rocprofv3 --kernel-trace --stats` run of this script (GPU-busy fraction of the launch-bound rollout loop)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dgppo_amd import _native as N
from dgppo_amd.engine import Engine, Hyper

dev = torch.device("cuda:0")
cfg = N.make_env_cfg(0, 8, 3)
B = int(os.environ.get("B", 4096))
eng = Engine(cfg, Hyper(), dev, use_graphs=os.environ.get("GRAPHS", "1") == "1")
seeds = torch.arange(B, device=dev, dtype=torch.int64)
reps = int(os.environ.get("REPS", 3))
eng.rollout(seeds, True, 1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for r in range(reps):
    eng.rollout(seeds, True, 2 + r)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f"rollout wall {dt * 1e3:.2f} ms  ({B * eng.T / dt / 1e6:.2f} M env-steps/s), total rollouts incl. warm-up: {reps + 1}")
