import sys, os, time
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/dgppo_amd") else os.getcwd())
import numpy as np, torch
from dgppo_amd import _native as N, engine as EN, init
dev = torch.device("cuda:0")
for kind, n, obs, kw in (("LidarBicycleTarget", 16, 8, {}), ("MPESpread", 3, 3, {}), ("LidarTarget", 5, 4, {}),
                         ("LidarSpread", 4, 2, {"n_rays": 16}),                 # generic env kernel (n_rays != 32)
                         ("LidarSpread", 6, 3, {"comm_radius": 15.0}),          # --full-observation: comm_radius = 10 * area
                         ("MPETarget", 1, 0, {}),                               # a single agent, no obstacles
                         # task variants: 2 / 1 landmark nodes, fixed obstacle layouts, the third (connectivity) cost
                         ("LidarLine", 6, 3, {}), ("MPELine", 3, 3, {}), ("MPELine", 6, 2, {}), ("MPEFormation", 8, 3, {}),
                         ("MPECorridor", 5, 2, {}), ("MPEConnectSpread", 6, 1, {})):
    cfg = N.make_env_cfg(N.ENV_KINDS[kind], n, obs, **kw)
    hp = EN.Hyper(batch_size=64 * 16, rnn_step=8, train_steps=10)
    eng = EN.Engine(cfg, hp, dev, T=16, use_graphs=True, multi_stream=True)
    eng.policy.load_tree(init.init_policy(0, cfg.node_dim, 2, hp.actor_gnn_layers))
    eng.Vl.load_tree(init.init_value(0, cfg.node_dim, 1, hp.Vl_gnn_layers, 2))
    eng.Vh.load_tree(init.init_value(0, cfg.node_dim, cfg.n_cost, hp.Vh_gnn_layers, 3))
    eng.set_entropy_noise(3)
    B = 128
    seeds = torch.arange(1, B + 1, device=dev, dtype=torch.int64) * 7919
    t0 = time.time()
    for it in range(3):
        ro, det = eng.rollout_pair(seeds + it, seeds + 1000 + it, noise_seed=it + 1)
        info = eng.update(ro, det, it, np.random.default_rng(it).permutation(B))
    torch.cuda.synchronize()
    ok = all(np.isfinite(v) for v in info.values())
    print(kind, n, obs, kw, "nodes", cfg.num_nodes, "ok" if ok else "NON-FINITE", {k: round(v, 4) for k, v in list(info.items())[:4]}, f"{time.time()-t0:.1f}s")
