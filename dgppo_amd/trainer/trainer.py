"""Training driver with the reference's loop order and logged keys (dgppo/trainer/trainer.py:20-141):
eval -> save -> collect -> update, wandb when importable else JSONL + stdout.

Data-parallel runs (one process per GPU): `rank` / `world` are two optional constructor arguments after the
reference's.  `n_env_train` stays the GLOBAL number of training environments: every rank draws the same global key array
and rolls out its contiguous share of it, the algo (built with the same `allreduce` / `world` / `rank`) exchanges
gradients once per minibatch, and evaluation, checkpoints and logging happen on rank 0 only — the parameters are
bit-identical on every rank, so nothing is lost."""
from __future__ import annotations

import json
import os
from time import time

import numpy as np
import torch


class _Logger:
    def __init__(self, log_dir, run_name, group, save_log):
        self.wandb = None
        self.path = os.path.join(log_dir, "metrics.jsonl") if save_log else None
        try:  # wandb is optional and must never touch the network here
            import wandb  # noqa: F401
            os.environ.setdefault("WANDB_MODE", "offline")
            wandb.init(name=run_name, project="dgppo", group=group, dir=log_dir)
            self.wandb = wandb
        except Exception:
            self.wandb = None

    def log(self, info: dict, step: int):
        if self.wandb is not None:
            self.wandb.log(info, step=step)
        if self.path is not None:
            with open(self.path, "a") as f:
                f.write(json.dumps({"step": step, **{k: float(v) for k, v in info.items()}}) + "\n")


class Trainer:
    def __init__(self, env, env_test, algo, gamma: float, n_env_train: int, n_env_test: int, log_dir: str, seed: int,
                 params: dict, save_log: bool = True, rank: int = 0, world: int = 1):
        self.env, self.env_test, self.algo, self.gamma = env, env_test, algo, gamma
        self.n_env_train, self.n_env_test, self.log_dir, self.seed = n_env_train, n_env_test, log_dir, seed
        self.rank, self.world = int(rank), int(world)
        assert 0 <= self.rank < self.world
        assert n_env_train % self.world == 0, f"n_env_train ({n_env_train}) must be a multiple of the number of ranks ({world})"
        assert getattr(algo, "world", 1) == self.world and getattr(algo, "rank", 0) == self.rank, \
            "the algo must be built with the same world / rank as the Trainer"
        save_log = save_log and self.rank == 0                   # one writer: rank 0
        if Trainer._check_params(params):
            self.params = params
        if save_log:
            os.makedirs(log_dir, exist_ok=True)
            self.model_dir = os.path.join(log_dir, "models")
            os.makedirs(self.model_dir, exist_ok=True)
        self.logger = _Logger(log_dir, params["run_name"], env.__class__.__name__, save_log) if self.rank == 0 else None
        self.save_log = save_log
        self.steps = params["training_steps"]
        self.eval_interval = params["eval_interval"]
        self.eval_epi = params["eval_epi"]
        self.save_interval = params["save_interval"]
        self.update_steps = 0
        self.key = np.random.default_rng([seed, 7])

    @staticmethod
    def _check_params(params: dict) -> bool:
        for k in ("run_name", "training_steps", "eval_interval", "eval_epi", "save_interval"):
            assert k in params, f"{k} not found in params"
        assert params["eval_interval"] > 0 and params["eval_epi"] >= 1 and params["save_interval"] > 0
        return True

    def evaluate(self, test_keys) -> dict:
        """reductions of trainer.py:105-125 on deterministic rollouts."""
        r = self.algo.collect_deterministic(test_keys, env=self.env_test)
        rewards = r.rewards.cpu().numpy()            # [E, T]
        costs = r.costs.cpu().numpy()                # [E, T, n, nh]
        total = rewards.sum(-1)
        return {
            "eval/reward": float(total.mean()), "eval/reward_final": float(rewards[:, -1].mean()),
            "eval/cost": float(np.maximum(costs, 0.0).max(-1).max(-1).sum(-1).mean()),
            "eval/unsafe_frac": float((costs.max(-1).max(-2) >= 1e-6).mean()),
            "_reward_min": float(total.min()), "_reward_max": float(total.max()),
        }

    def train(self):
        start_time = time()
        assert self.n_env_test <= 1000, "n_env_test must be less than or equal to 1_000"
        test_keys = np.random.default_rng([self.seed, 11]).integers(1, 2 ** 62, size=1000)[:self.n_env_test]
        for step in range(0, self.steps + 1):
            if step % self.eval_interval == 0 and self.rank == 0:
                ev = self.evaluate(test_keys)
                rmin, rmax = ev.pop("_reward_min"), ev.pop("_reward_max")
                print(f"step: {step:3}, time: {time() - start_time:5.0f}s, reward: {ev['eval/reward']:9.4f}, "
                      f"min/max reward: {rmin:7.2f}/{rmax:7.2f}, cost: {ev['eval/cost']:8.4f}, "
                      f"unsafe_frac: {ev['eval/unsafe_frac']:6.2f}", flush=True)
                self.logger.log(ev, step=self.update_steps)
            if self.save_log and step % self.save_interval == 0:
                self.algo.save(os.path.join(self.model_dir), step)
            keys = self.key.integers(1, 2 ** 62, size=self.n_env_train)        # the global batch, identical on every rank
            share = self.n_env_train // self.world
            rollouts = self.algo.collect(None, keys[self.rank * share:(self.rank + 1) * share])
            update_info = self.algo.update(rollouts, step)                      # global values on every rank (Engine.info)
            if self.logger is not None:
                self.logger.log(update_info, step=self.update_steps)
            self.update_steps += 1
