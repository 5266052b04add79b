"""Evaluation statistics of the reference's test.py (test.py:103-141): per-episode reward / cost / safe rate and the
aggregate safe-rate mean and standard deviation.  Host reductions on the arrays a test rollout returns.

`unsafe_mask` follows test.py:103-105 — `any(env.get_cost(graph) >= 0, axis=-1)` on the T pre-step graphs of a rollout —
and `Rollout.costs[t]` IS `get_cost(graph[t])` (cost is evaluated on the pre-step graph, lidar_env/base.py:170-171), so the
mask is taken from the stored costs instead of re-deriving the graphs."""
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict, List

import numpy as np
import yaml


def unsafe_mask(costs: np.ndarray) -> np.ndarray:
    """costs [..., n_agents, n_cost] -> bool [..., n_agents]  (test.py:103-105)."""
    return np.any(np.asarray(costs) >= 0.0, axis=-1)


def episode_stats(rewards: np.ndarray, costs: np.ndarray) -> Dict[str, np.ndarray]:
    """rewards [E, T], costs [E, T, n, nh] -> per-episode reward sum, cost max, safe rate (test.py:118-127) and the
    per-(episode, agent) unsafe flag `max_t is_unsafe` used by the aggregate (test.py:131)."""
    rewards, costs = np.asarray(rewards), np.asarray(costs)
    is_unsafe = unsafe_mask(costs)                               # [E, T, n]
    ever_unsafe = is_unsafe.max(axis=1)                          # [E, n]
    return {
        "reward": rewards.sum(axis=1),
        "cost": costs.reshape(costs.shape[0], -1).max(axis=1),
        "safe_rate": 1.0 - ever_unsafe.mean(axis=1),
        "ever_unsafe": ever_unsafe,
    }


def aggregate(stats: Dict[str, np.ndarray]) -> Dict[str, float]:
    """test.py:131-138: safe mean / std over all (episode, agent) pairs, reward and cost mean / min / max."""
    safe = 1.0 - stats["ever_unsafe"].astype(np.float64)
    r, c = stats["reward"], stats["cost"]
    return {"reward": float(r.mean()), "reward_min": float(r.min()), "reward_max": float(r.max()),
            "cost": float(c.mean()), "cost_min": float(c.min()), "cost_max": float(c.max()),
            "safe_mean": float(safe.mean()), "safe_std": float(safe.std())}


def csv_line(env, epi: int, agg: Dict[str, float]) -> str:
    """the row test.py:142-146 appends to test_log.csv"""
    return (f"{env.num_agents},{epi},{env.max_episode_steps},{env.area_size},{env.params['n_obs']},"
            f"{agg['safe_mean'] * 100:.3f},{agg['safe_std'] * 100:.3f}\n")


# ---- config.yaml --------------------------------------------------------------------------------------------------
class _ConfigLoader(yaml.SafeLoader):
    """SafeLoader that also understands the one python tag the reference's train.py writes (`yaml.dump(args)` of an
    argparse.Namespace, train.py:119-121) by reading it as a plain mapping.  Nothing from the file is executed."""


_ConfigLoader.add_constructor("tag:yaml.org,2002:python/object:argparse.Namespace",
                              lambda loader, node: loader.construct_mapping(node, deep=True))


def load_config(path: str) -> SimpleNamespace:
    """{log_dir}/config.yaml -> attribute access like the reference's `config.env`, `config.num_agents`, ...
    (test.py:36-38 uses yaml.UnsafeLoader; this loader refuses every other python tag)."""
    with open(path, "r") as f:
        data = yaml.load(f, Loader=_ConfigLoader)
    if not isinstance(data, dict):
        raise ValueError(f"{path}: expected a mapping, got {type(data).__name__}")
    return SimpleNamespace(**data)


def latest_step(model_path: str) -> int:
    """test.py:53-55: the largest all-digit directory name under models/"""
    import os
    steps: List[int] = [int(m) for m in os.listdir(model_path) if m.isdigit()]
    if not steps:
        raise FileNotFoundError(f"no checkpoints under {model_path}")
    return max(steps)
