#!/usr/bin/env python
"""Rehearsal of the data-parallel update on ONE GPU: `world` ranks share the device and exchange gradients over gloo (RCCL
refuses two ranks on one device), each driving Engine(allreduce=..., world=...) exactly as bench.py / train.py do.

Every rank collects the SAME 2*world environments (same seeds, same noise), so the targets are identical everywhere; rank r
then puts "its" shard of the minibatch first in the permutation.  After the all-reduce the summed gradient / world must
equal the gradient a single process gets on the whole minibatch (tests/test_engine_gpu.py::test_two_rank_update...), and
the replicas' parameters must stay bit-identical after all optimiser steps.

    python tools/dist_rehearsal.py --rank R --world W --port P --out FILE      (one process per rank)
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build_engine(device, world, allreduce, T_=8, rs=4, shard_envs=2, multi_stream=True):
    from dgppo_amd import _native as N, engine as EN, init
    from oracle import nn_torch as T
    cfg = N.make_env_cfg(N.ENV_KINDS["LidarSpread"], 3, 2)
    hp = EN.Hyper(batch_size=shard_envs * T_, rnn_step=rs, train_steps=100)      # per-rank minibatch = shard_envs envs
    eng = EN.Engine(cfg, hp, device, T=T_, allreduce=allreduce, world=world, multi_stream=multi_stream)
    trees = {"policy": init.init_policy(0, cfg.node_dim, 2, 2), "Vl": init.init_value(0, cfg.node_dim, 1, 2, 2),
             "Vh": init.init_value(0, cfg.node_dim, 2, 1, 3)}
    rng = np.random.default_rng(11)
    trees = {k: T.tree_map(lambda a: torch.from_numpy(a + 0.05 * rng.standard_normal(a.shape).astype(np.float32)), v)
             for k, v in trees.items()}
    for k, net in eng.nets.items():
        net.load_tree(trees[k])
    eng.set_entropy_noise(77)
    return eng


def run(eng, device, B, perm, first_only=False):
    """collect B envs, update with `perm`; returns the gradients seen by the optimiser at minibatch 0 and the parameters."""
    seeds = torch.arange(1, B + 1, dtype=torch.int64, device=device) * 7919
    ro = eng.rollout(seeds, True, noise_seed=3)
    det = eng.rollout(seeds + 1000, False)
    grads = {}

    def hook(name, net, mb):
        if mb == 0:
            grads[name] = net.grads.detach().clone()
    eng.grad_hook = hook
    info = eng.update(ro, det, 10, np.asarray(perm))
    torch.cuda.synchronize()
    return ({k: v.cpu() for k, v in grads.items()}, {k: net.params.detach().cpu().clone() for k, net in eng.nets.items()},
            {k: float(v) for k, v in info.items()})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rank", type=int, required=True)
    ap.add_argument("--world", type=int, default=2)
    ap.add_argument("--port", type=int, required=True)
    ap.add_argument("--out", type=str, required=True)
    a = ap.parse_args()
    os.environ.update(RANK=str(a.rank), LOCAL_RANK=str(a.rank), WORLD_SIZE=str(a.world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(a.port))
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    from dgppo_amd import dist as D
    D.init_control_plane(timeout_s=300)
    allreduce, close = D.make_allreduce(a.world, backend="gloo")
    eng = build_engine(device, a.world, allreduce)
    B = 2 * a.world
    # rank r: its own shard (envs 2r, 2r+1) first, then the next rank's, ... : every minibatch step the ranks' shards are
    # disjoint and together cover the same global minibatch a single process takes with batch_size * world
    perm = [(2 * ((a.rank + j) % a.world)) + i for j in range(a.world) for i in range(2)]
    grads, params, info = run(eng, device, B, perm)
    torch.save({"grads": grads, "params": params, "info": info, "perm": perm}, a.out)
    D.barrier(a.world)
    close()
    D.shutdown(a.world)


if __name__ == "__main__":
    main()
