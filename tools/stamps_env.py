"""Developer tool: per-phase cycle stamps of the LiDAR step kernel (needs a -DDGPPO_STAMPS build of env_step.o)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dgppo_amd import _native as N, ops_env as OE
dev = torch.device("cuda:0")
cfg = N.make_env_cfg(0, 8, 3)
for B in (1, 256, 4096):
    seeds = torch.arange(1, B + 1, dtype=torch.int64, device=dev) * 7919
    agent = torch.empty(B, 8, 4, device=dev); goal = torch.empty(B, 8, 4, device=dev); obst = torch.empty(B, 3, 16, device=dev)
    OE.env_reset(cfg, seeds, agent, goal, obst)
    rc, rs = OE.ray_tables(32, dev)
    hits = torch.empty(B, 8, 8, 2, device=dev)
    OE.env_step(cfg, agent, None, goal, obst, None, rc, rs, None, hits, None, None, None)
    act = torch.empty(B, 8, 2, device=dev).uniform_(-1, 1)
    nx = torch.empty_like(agent); nh = torch.empty_like(hits); rew = torch.empty(B, device=dev); cost = torch.empty(B, 8, 2, device=dev)
    g = OE.alloc_graph(cfg, B, dev)
    for _ in range(3):
        OE.env_step(cfg, agent, act, goal, obst, hits, rc, rs, nx, nh, rew, cost, g)
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 32)()
    N.lib().dgppo_debug_stamps(out)
    st = np.array(out[:10], dtype=np.int64)
    print("B", B, "block", os.environ.get("DGPPO_ENV_BLOCK", "64"), "phase ticks (100MHz clock? s_memtime):", (st[1:] - st[:-1]).tolist(), "total", int(st[9] - st[0]))
