"""Developer tool: the update phase alone (value pre-passes + 2 GAE + advantage + 32 x (Vl, Vh, policy) minibatch updates)
of the benchmark workload, repeated on ONE collected rollout pair, so that a `rocprofv3 --kernel-trace --stats` run of this
script is dominated by update kernels:   rocprofv3 --kernel-trace --stats --output-format csv -d OUT -- python3 tools/prof_update_only.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dgppo_amd import _native as N, init
from dgppo_amd.engine import Engine, Hyper

dev = torch.device("cuda:0")
cfg = N.make_env_cfg(0, 8, 3)
B = int(os.environ.get("B", "4096"))
REPS = int(os.environ.get("REPS", "4"))
hp = Hyper(batch_size=16384, train_steps=1000)
eng = Engine(cfg, hp, dev, use_graphs=False, multi_stream=os.environ.get("MS", "1") == "1")
eng.policy.load_tree(init.init_policy(0, cfg.node_dim, 2, hp.actor_gnn_layers))
eng.Vl.load_tree(init.init_value(0, cfg.node_dim, 1, hp.Vl_gnn_layers, 2))
eng.Vh.load_tree(init.init_value(0, cfg.node_dim, 2, hp.Vh_gnn_layers, 3))
eng.set_entropy_noise(1)
seeds = torch.arange(1, B + 1, device=dev, dtype=torch.int64)
rng = np.random.default_rng(0)
ro, det = eng.rollout_pair(seeds, seeds + 100000, noise_seed=1)
ro.finalize(); det.finalize()
torch.cuda.synchronize()
for it in range(REPS):
    t0 = time.perf_counter()
    tg = eng.targets(ro, det, it)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    orig = eng.targets
    eng.targets = lambda *a, **k: tg
    eng.update(ro, det, it, rng.permutation(B))
    torch.cuda.synchronize(); t2 = time.perf_counter()
    eng.targets = orig
    print(f"rep {it}: targets {1e3 * (t1 - t0):.1f} ms, minibatch loop {1e3 * (t2 - t1):.1f} ms", flush=True)
