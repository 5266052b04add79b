"""Generates tests/golden/*.npz from the ORACLE (the reference itself is not importable here — SURVEY F3 — so these vectors
pin the oracle against accidental edits and give the HIP kernels fixed inputs/outputs that travel to the GPU box).
Run from the repo root:  python tests/golden/make_golden.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import algo_ref as A, env_np as E, nn_torch as T  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def env_case(name, kind, n, n_obs, B, seed):
    ocfg = E.EnvCfg(kind, n_agents=n, n_obs=n_obs)
    rng = np.random.default_rng(seed)
    agent, goal, obst = E.env_reset(ocfg, rng.integers(1, 2 ** 60, size=B))
    agent[:, :, :2] = (agent[:, :, :2] * 0.5 + 0.4).astype(np.float32)
    agent[:, :, 2:4] = rng.uniform(-0.5, 0.5, size=(B, n, 2)).astype(np.float32)
    tab = E.ray_table(32)
    hits = E.lidar_sense(ocfg, agent[..., :2], obst, *tab)[0] if (ocfg.is_lidar and n_obs > 0) else None
    action = rng.uniform(-1.4, 1.4, size=(B, n, 2)).astype(np.float32)
    out = E.env_step(ocfg, agent, goal, obst, hits, action, tab)
    d = dict(kind=kind, n=n, n_obs=n_obs, agent=agent, goal=goal, action=action, next_agent=out["next_agent"],
             reward=out["reward"], cost=out["cost"], **{"g_" + k: v for k, v in out["graph"].items()})
    if obst is not None and obst.shape[1] > 0:
        d["obst"] = obst
    if hits is not None:
        d["hits"], d["next_hits"] = hits, out["next_hits"]
    np.savez_compressed(os.path.join(OUT, name), **d)


def gae_case():
    r = np.random.default_rng(3)
    B, T_, n, nh = 3, 16, 3, 2
    costs = r.uniform(-1, 1, size=(B, T_, n, nh)).astype(np.float32)
    rew = (-r.uniform(0, 0.02, size=(B, T_))).astype(np.float32)
    Vh = r.uniform(-1, 1, size=(B, T_ + 1, n, nh)).astype(np.float32)
    Vl = r.uniform(0, 1, size=(B, T_ + 1)).astype(np.float32)
    Qh, Ql = A.gae_batch(costs, rew, Vh, Vl, 0.99, 0.95)
    adv, safe = A.advantage(Ql, Vl, (Vh * 0.03).astype(np.float32), 0.03, 10.0, 1e-2, 2.0)
    np.savez_compressed(os.path.join(OUT, "gae.npz"), costs=costs, rewards=rew, Vh=Vh, Vl=Vl, Qh=Qh, Ql=Ql, adv=adv, safe=safe)


def policy_case():
    ocfg = E.EnvCfg(E.LIDAR_SPREAD, n_agents=3, n_obs=2)
    rng = np.random.default_rng(5)
    B = 4
    agent, goal, obst = E.env_reset(ocfg, rng.integers(1, 2 ** 60, size=B))
    agent[:, :, :2] = (agent[:, :, :2] * 0.5 + 0.4).astype(np.float32)
    hits, _ = E.lidar_sense(ocfg, agent[..., :2], obst, *E.ray_table(32))
    tree = T.init_policy(0, 7)
    tree["params"]["ScaleHid"]["kernel"] = T.orthogonal(torch.Generator().manual_seed(1), 64, 64, 0.5)
    h = torch.from_numpy(rng.standard_normal((B, 3, 64)).astype(np.float32) * 0.3)
    eps = torch.from_numpy(rng.standard_normal((B, 3, 2)).astype(np.float32))
    g = T.graph_to_torch(E.get_graph(ocfg, agent, goal, obst, hits))
    with torch.no_grad():
        a, lp, h_new = T.policy_sample(tree, g, h, 3, eps)
        mode, _ = T.policy_mode(tree, g, h, 3)
    leaves = {"p" + k.replace("/", "."): v.numpy() for k, v in T.tree_leaves(tree)}
    np.savez_compressed(os.path.join(OUT, "policy_lidar_spread_n3.npz"), agent=agent, goal=goal, obst=obst, hits=hits,
                        h=h.numpy(), eps=eps.numpy(), action=a.numpy(), log_pi=lp.numpy(), h_new=h_new.numpy(),
                        mode=mode.numpy(), **leaves)


if __name__ == "__main__":
    env_case("env_lidar_spread_n3.npz", E.LIDAR_SPREAD, 3, 2, 4, 1)
    env_case("env_mpe_target_n3.npz", E.MPE_TARGET, 3, 0, 4, 2)
    env_case("env_lidar_target_n4.npz", E.LIDAR_TARGET, 4, 3, 3, 3)
    gae_case()
    policy_case()
    print(sorted(os.listdir(OUT)))
