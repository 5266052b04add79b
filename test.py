#!/usr/bin/env python
"""Drop-in for the reference's test.py (same flags and defaults, test.py:159-189) on the MI355X implementation: load
`{path}/config.yaml` and `{path}/models/{step}/{actor,Vl,Vh}.pkl`, run `--epi` test episodes with the deterministic
(or `--stochastic`) policy, print per-episode reward / cost / safe rate and the aggregate, optionally append
`test_log.csv`.  All episodes run as ONE batched rollout on the GPU (the reference loops over episodes on the host).
Unless `--no-video`, every episode is rendered to `{path}/videos/{step}/` (test.py:150-159; `.gif` when no ffmpeg binary
is available for `.mp4`)."""
import argparse
import datetime
import os
import pathlib

import numpy as np

from dgppo.algo import make_algo
from dgppo.env import make_env
from dgppo_amd.trainer import evaluate as EV


def test(args):
    print(f"> Running test.py {args}")
    np.random.seed(args.seed)
    config = EV.load_config(os.path.join(args.path, "config.yaml"))
    num_agents = config.num_agents if args.num_agents is None else args.num_agents
    env = make_env(env_id=config.env if args.env is None else args.env, num_agents=num_agents,
                   num_obs=config.obs if args.obs is None else args.obs, max_step=args.max_step,
                   full_observation=args.full_observation)
    model_path = os.path.join(args.path, "models")
    step = EV.latest_step(model_path) if args.step is None else args.step
    print("step: ", step)
    algo = make_algo(
        algo=config.algo, env=env, node_dim=env.node_dim, edge_dim=env.edge_dim, state_dim=env.state_dim,
        action_dim=env.action_dim, n_agents=env.num_agents, cost_weight=config.cost_weight,
        actor_gnn_layers=config.actor_gnn_layers, Vl_gnn_layers=config.Vl_gnn_layers,
        Vh_gnn_layers=getattr(config, "Vh_gnn_layers", 1), lr_actor=config.lr_actor, lr_Vl=config.lr_Vl, max_grad_norm=2.0,
        seed=config.seed, use_rnn=config.use_rnn, rnn_layers=config.rnn_layers, use_lstm=config.use_lstm)
    algo.load(model_path, step)

    # episode seeds: the reference takes jr.split(PRNGKey(seed), 1000)[:epi][offset:] (test.py:90-93); RNG streams are not
    # comparable across libraries (threefry vs Philox), the structure (1000 draws, prefix, offset) is kept
    keys = np.random.default_rng([args.seed, 13]).integers(1, 2 ** 62, size=1000)[:args.epi][args.offset:]
    if len(keys) == 0:
        raise SystemExit("no episodes to run (--epi / --offset)")
    if args.stochastic:
        ro = algo.collect_stochastic(keys, env=env)
    else:
        ro = algo.collect_deterministic(keys, env=env)
    stats = EV.episode_stats(ro.rewards.cpu().numpy(), ro.costs.cpu().numpy())
    for i in range(len(keys)):
        print(f"epi: {i}, reward: {stats['reward'][i]:.3f}, cost: {stats['cost'][i]:.3f}, "
              f"safe rate: {stats['safe_rate'][i] * 100:.3f}%")
    agg = EV.aggregate(stats)
    print(f"reward: {agg['reward']:.3f}, min/max reward: {agg['reward_min']:.3f}/{agg['reward_max']:.3f}, "
          f"cost: {agg['cost']:.3f}, min/max cost: {agg['cost_min']:.3f}/{agg['cost_max']:.3f}, "
          f"safe_rate: {agg['safe_mean'] * 100:.3f}%")
    if args.log:
        with open(os.path.join(args.path, "test_log.csv"), "a") as f:
            f.write(EV.csv_line(env, args.epi, agg))
    if not args.no_video:
        videos_dir = pathlib.Path(args.path) / "videos" / f"{step}"
        videos_dir.mkdir(exist_ok=True, parents=True)
        stamp = datetime.datetime.now().strftime("%m%d-%H%M")
        unsafe = (ro.costs >= 0.0).any(dim=-1).cpu().numpy()          # [B, T, n]  (test.py:103-105)
        for i in range(len(keys)):
            name = (f"n{num_agents}_epi{i:02}_reward{stats['reward'][i]:.3f}_cost{stats['cost'][i]:.3f}"
                    f"_sr{stats['safe_rate'][i] * 100:.0f}")
            out = env.render_video(ro, videos_dir / f"{stamp}_{name}.mp4", unsafe[i], {}, dpi=args.dpi, index=i)
            print(f"video: {out}")
    return agg


# the reference's flags (test.py:159-189): (names, kind, default); kind is a type name or "flag"
FLAGS = [
    (("--path",), "req:str", None), (("--no-video",), "flag", False), (("--epi",), "int", 5), (("--step",), "int", None),
    (("--obs",), "int", None), (("--stochastic",), "flag", False), (("--full-observation",), "flag", False),
    (("--debug",), "flag", False), (("--cpu",), "flag", False), (("--max-step",), "int", None), (("--log",), "flag", False),
    (("-n", "--num-agents"), "int", None), (("--seed",), "int", 1234), (("--env",), "str", None), (("--offset",), "int", 0),
    (("--dpi",), "int", 100),
]


def main(argv=None):
    types = {"int": int, "str": str}
    ap = argparse.ArgumentParser(description=__doc__)
    for names, kind, default in FLAGS:
        if kind == "flag":
            ap.add_argument(*names, action="store_true", default=False)
        elif kind.startswith("req:"):
            ap.add_argument(*names, type=types[kind[4:]], required=True)
        else:
            ap.add_argument(*names, type=types[kind], default=default)
    args = ap.parse_args(argv)
    if args.cpu:
        raise SystemExit("--cpu: this build has no CPU product path (the HIP library is required)")
    return test(args)


if __name__ == "__main__":
    main()
