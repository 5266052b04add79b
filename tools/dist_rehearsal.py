#!/usr/bin/env python
"""Rehearsal of data-parallel training on ONE GPU: `world` ranks share the device and exchange gradients over gloo (RCCL
refuses two ranks on one device), each driving the engine exactly as bench.py / train.py do.  Launched as supervised
ranks by dgppo_amd/launch.py (RANK / WORLD_SIZE / DGPPO_RDZV_FILE in the environment); `--world 1` runs the single-process
twin on the UNION batch in the caller's process.

The sharding is the real one: rank r owns the global envs [r * B_local, (r + 1) * B_local) — scene seeds and
sampling noise are functions of the global env index (Engine(rank=r): dgppo_randn_rows) — and every rank walks the same
permutation of its local env indices, so minibatch k of the job is the union of the ranks' k-th slices.  The
single-process twin takes the union batch with the union permutation; what must hold:
  * the all-reduced gradient / world equals the twin's gradient on the union minibatch (mean of equal shards = global mean);
  * replicas hold bit-identical parameters after every optimiser step;
  * the parameters follow the twin's within fp32 summation-order noise.

modes:
  eager    Engine(use_graphs=False) + a gradient hook (first-minibatch gradients are saved), one iteration
  graphs   Engine(use_graphs=True): graph(body_pre) -> eager all-reduce -> graph(optimiser), 3 iterations so that the later
           ones replay; parameters after every iteration are saved (this path had no world > 1 coverage)
  trainer  the drop-in path: make_env / make_algo(allreduce, world, rank) / Trainer(rank, world).train() for 2 iterations
  lagr     graphs mode with algo = informarl_lagr (multiplier sums all-reduced)

    python tools/dist_rehearsal.py --mode M --out PREFIX           (one process per rank; writes PREFIX.r{rank}.pt)
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

T_, RS, B_LOCAL, EB_LOCAL = 8, 4, 4, 2          # horizon, rnn_step, envs per rank, envs per rank per minibatch (world = 2)


def union_perm(local_perm, world, B_local, Eb_local):
    """the single-process permutation whose k-th minibatch (world * Eb_local envs) is the union of the ranks' k-th slices"""
    out = []
    for k in range(B_local // Eb_local):
        sl = np.asarray(local_perm[k * Eb_local:(k + 1) * Eb_local])
        for r in range(world):
            out.extend((sl + r * B_local).tolist())
    return np.asarray(out)


def build_engine(device, world, rank, allreduce, share, use_graphs=False, multi_stream=True, algo="dgppo"):
    """share = how many ranks' worth of envs THIS engine's minibatch holds (1 for a rank, world for the twin)"""
    from dgppo_amd import _native as N, engine as EN, init
    from oracle import nn_torch as T
    cfg = N.make_env_cfg(N.ENV_KINDS["LidarSpread"], 3, 2)
    hp = EN.Hyper(batch_size=share * EB_LOCAL * T_, rnn_step=RS, train_steps=100, lr_lagr=1e-2)
    eng = EN.Engine(cfg, hp, device, T=T_, allreduce=allreduce, world=world, rank=rank, use_graphs=use_graphs,
                    multi_stream=multi_stream, algo=algo)
    trees = {"policy": init.init_policy(0, cfg.node_dim, 2, 2), "Vl": init.init_value(0, cfg.node_dim, 1, 2, 2),
             "Vh": init.init_value(0, cfg.node_dim, 2, 1, 3, global_info=(algo == "informarl_lagr"))}
    rng = np.random.default_rng(11)
    trees = {k: T.tree_map(lambda a: torch.from_numpy(a + 0.05 * rng.standard_normal(a.shape).astype(np.float32)), v)
             for k, v in trees.items()}
    for k, net in eng.nets.items():
        net.load_tree(trees[k])
    eng.set_entropy_noise(77)
    return eng


def global_seeds(it, n_global):
    return (np.arange(1, n_global + 1, dtype=np.int64) + 50 * it) * 7919


def run_engine(eng, device, world_total, rank, iters, hook):
    """`iters` iterations on this engine's share of the global batch; world_total = number of shards the global batch has"""
    n_global = world_total * B_LOCAL
    share = n_global // eng.world                                  # envs of this engine
    grads, params_per_iter, infos = {}, [], []
    if hook:
        def h(name, net, mb):
            if mb == 0 and name not in grads:
                grads[name] = net.grads.detach().clone()
        eng.grad_hook = h
    for it in range(iters):
        gs = global_seeds(it, n_global)
        mine = torch.from_numpy(gs[rank * share:(rank + 1) * share]).to(device)
        ro = eng.rollout(mine, True, noise_seed=3 + it)
        det = eng.rollout(mine + 1000, False) if eng.algo == "dgppo" else None
        local = np.random.default_rng([5, it]).permutation(B_LOCAL)   # the same on every rank
        perm = local if eng.world == world_total else union_perm(local, world_total, B_LOCAL, EB_LOCAL)
        infos.append({k: float(v) for k, v in eng.update(ro, det, 10 + it, perm).items()})
        torch.cuda.synchronize()
        params_per_iter.append({k: net.params.detach().cpu().clone() for k, net in eng.nets.items()})
    out = {"grads": {k: v.cpu() for k, v in grads.items()}, "params": params_per_iter, "info": infos}
    if eng.algo == "informarl_lagr":
        out["lagr"] = eng.lagr.detach().cpu().clone()
    return out


def run_trainer(device, world, rank, allreduce, world_total, log_dir):
    """2 training iterations through the public API (make_env / make_algo / Trainer), as train.py wires them"""
    from dgppo.algo import make_algo
    from dgppo.env import make_env
    from dgppo.trainer.trainer import Trainer
    np.random.seed(0)
    n_global = world_total * B_LOCAL
    env = make_env("LidarSpread", 3, max_step=T_, num_obs=2)
    env_test = make_env("LidarSpread", 3, max_step=T_, num_obs=2)
    algo = make_algo("dgppo", env=env, node_dim=env.node_dim, edge_dim=env.edge_dim, state_dim=env.state_dim,
                     action_dim=env.action_dim, n_agents=env.num_agents, seed=0, train_steps=100,
                     batch_size=(world_total // world) * EB_LOCAL * T_, rnn_step=RS, allreduce=allreduce, world=world, rank=rank)
    if world != world_total:                                       # the twin: union permutation from the ranks' generator
        g = np.random.default_rng([0, 4242])
        algo.perm_fn = lambda B: union_perm(g.permutation(B_LOCAL), world_total, B_LOCAL, EB_LOCAL)
    sched = {"run_name": "rehearsal", "training_steps": 1, "eval_interval": 1, "eval_epi": 1, "save_interval": 1}
    tr = Trainer(env=env, env_test=env_test, algo=algo, gamma=0.99, n_env_train=n_global, n_env_test=2, log_dir=log_dir,
                 seed=0, params=sched, save_log=True, rank=rank, world=world)
    tr.train()                                                     # steps 0 and 1: two collect + update iterations
    torch.cuda.synchronize()
    return {"params": [{k: net.params.detach().cpu().clone() for k, net in algo.engine.nets.items()}],
            "wrote_logs": os.path.exists(os.path.join(log_dir, "metrics.jsonl")),
            "saved_models": sorted(os.listdir(os.path.join(log_dir, "models"))) if os.path.isdir(os.path.join(log_dir, "models")) else []}


def run_mode(mode, device, world, rank, allreduce, world_total, log_dir=None):
    if mode == "trainer":
        return run_trainer(device, world, rank, allreduce, world_total, log_dir)
    algo = "informarl_lagr" if mode == "lagr" else "dgppo"
    eng = build_engine(device, world, rank, allreduce, share=world_total // world, use_graphs=(mode != "eager"), algo=algo)
    return run_engine(eng, device, world_total, rank, iters=1 if mode == "eager" else 3, hook=(mode == "eager"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", choices=("eager", "graphs", "trainer", "lagr"), required=True)
    ap.add_argument("--out", type=str, required=True)
    a = ap.parse_args()
    from dgppo_amd import dist as D
    rank, _, world = D.env_info()
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    D.init_control_plane(timeout_s=300)
    allreduce, close = D.make_allreduce(world, backend="gloo")
    D.selfcheck_allreduce(allreduce, rank, world, device)
    res = run_mode(a.mode, device, world, rank, allreduce, world, log_dir=f"{a.out}.logs.r{rank}")
    torch.save(res, f"{a.out}.r{rank}.pt")
    print(f"rank {rank}: mode {a.mode} done", file=sys.stderr, flush=True)
    D.barrier(world)
    close()
    D.shutdown(world)


if __name__ == "__main__":
    main()
