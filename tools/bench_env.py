"""Micro-benchmark + quick parity of the raycast+graph kernel (developer tool; same loop as bench.py's roofline part)."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from dgppo_amd import _native as N, ops_env as OE  # noqa: E402
from oracle import env_np as E  # noqa: E402

dev = torch.device("cuda:0")
cfg = N.make_env_cfg(0, 8, 3)
# parity on 512 envs, 2 chained steps, everything bit-exact
ocfg = E.EnvCfg(0, n_agents=8, n_obs=3)
B = 512
seeds = torch.arange(1, B + 1, dtype=torch.int64, device=dev) * 7919
agent = torch.empty(B, 8, 4, device=dev); goal = torch.empty(B, 8, 4, device=dev); obst = torch.empty(B, 3, 16, device=dev)
OE.env_reset(cfg, seeds, agent, goal, obst)
rc, rs = OE.ray_tables(32, dev)
hits = torch.empty(B, 8, 8, 2, device=dev)
OE.env_step(cfg, agent, None, goal, obst, None, rc, rs, None, hits, None, None, None)
a_np, g_np, o_np, h_np = (x.cpu().numpy() for x in (agent, goal, obst, hits))
rng = np.random.default_rng(0)
ok = True
for t in range(2):
    act = rng.uniform(-1.2, 1.2, size=(B, 8, 2)).astype(np.float32)
    nx = torch.empty_like(agent); nh = torch.empty_like(hits); rew = torch.empty(B, device=dev); cost = torch.empty(B, 8, 2, device=dev)
    g = OE.alloc_graph(cfg, B, dev)
    t_ = lambda x: torch.from_numpy(x).to(dev)
    OE.env_step(cfg, t_(a_np), t_(act), t_(g_np), t_(o_np), t_(h_np), rc, rs, nx, nh, rew, cost, g)
    want = E.env_step(ocfg, a_np, g_np, o_np, h_np, act, E.ray_table(32))
    for k, v in (("next_agent", nx), ("next_hits", nh), ("reward", rew), ("cost", cost)):
        ok &= np.array_equal(v.cpu().numpy().view(np.uint32), want[k].view(np.uint32))
    for k, v in g.items():
        ok &= np.array_equal(v.cpu().numpy(), want["graph"][k])
    a_np, h_np = nx.cpu().numpy(), nh.cpu().numpy()
print("parity bit-exact:", ok)
for Bn in ([int(x) for x in os.environ["SIZES"].split(",")] if "SIZES" in os.environ else (4096, 16384)):
    r = bench.roofline_env_kernel(cfg, dev, Bn, iters=100)
    print(Bn, json.dumps({k: {kk: round(vv, 2) for kk, vv in v.items()} for k, v in r.items()}))
