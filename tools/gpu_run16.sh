#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_algo_gpu.py -x -q -m gpu > $O/r3j_t.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -12 $O/r3j_t.log | cut -c1-250
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python3 - <<'PY'
import torch, time, os, sys
sys.path.insert(0, os.getcwd())
from dgppo_amd import ops_algo as O
dev = torch.device("cuda:0")
B, T, n, nh = 4096, 128, 8, 2
costs = torch.rand(B, T, n, nh, device=dev) * 2 - 1; rew = -torch.rand(B, T, device=dev) * 0.02
Vh = torch.rand(B, T + 1, n, nh, device=dev) * 2 - 1; Vl = torch.rand(B, T + 1, device=dev)
Qh = torch.empty(B, T, n, nh, device=dev); Ql = torch.empty(B, T, device=dev)
lp = O.lam_pow_table(0.95, T, dev)
for env in ({}, {"DGPPO_GAE_ROWS": "1"}):
    os.environ.update(env)
    for _ in range(3): O.gae(costs, rew, Vh, Vl, lp, 0.99, 0.95, Qh, Ql)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): O.gae(costs, rew, Vh, Vl, lp, 0.99, 0.95, Qh, Ql)
    torch.cuda.synchronize(); print(env, (time.perf_counter() - t0) / 10 * 1e6, "us")
PY
