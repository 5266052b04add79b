"""InforMARL (the reference's PPO baseline, dgppo/algo/informarl.py:27-457) on the same HIP engine: graph-transformer
actor + cost critic Vl, no constraint-value network.  The stage cost is -reward + cost_weight * sum(max(cost, 0))
(informarl.py:329), the advantage the per-env normalised -(Ql - Vl) (informarl.py:334-336); Vl and policy updates are the
ones DGPPO inherits (informarl.py:357-457).  SURVEY §8f rank 3."""
from __future__ import annotations

import os

import numpy as np
import torch

from .. import engine as EN
from .. import init as INIT
from .. import nets
from ..utils import checkpoint as CK
from .base import Algorithm
from .dgppo import DGPPO, _check_rnn_options, _n_cells


class InforMARL(DGPPO):
    def __init__(self, env, node_dim: int, edge_dim: int, state_dim: int, action_dim: int, n_agents: int,
                 cost_weight: float = 0.0, actor_gnn_layers: int = 2, Vl_gnn_layers: int = 2, gamma: float = 0.99,
                 lr_actor: float = 3e-4, lr_Vl: float = 1e-3, batch_size: int = 8192, epoch_ppo: int = 1,
                 clip_eps: float = 0.25, gae_lambda: float = 0.95, coef_ent: float = 1e-2, max_grad_norm: float = 2.0,
                 seed: int = 0, use_rnn: bool = True, rnn_layers: int = 1, rnn_step: int = 16, use_lstm: bool = False,
                 cost_schedule: bool = False, train_steps: int = 1e5, allreduce=None, world: int = 1, rank: int = 0, **kwargs):
        Algorithm.__init__(self, env, node_dim, edge_dim, action_dim, n_agents)
        _check_rnn_options(use_rnn, use_lstm, rnn_layers)
        assert epoch_ppo >= 1
        assert node_dim == env.node_dim and action_dim == 2
        self.state_dim, self.seed = state_dim, seed
        self.epoch_ppo, self.use_rnn, self.rnn_layers, self.use_lstm = epoch_ppo, use_rnn, rnn_layers, use_lstm
        self.hp = EN.Hyper(gamma=gamma, gae_lambda=gae_lambda, clip_eps=clip_eps, coef_ent=coef_ent,
                           max_grad_norm=max_grad_norm, lr_actor=lr_actor, lr_Vl=lr_Vl, batch_size=batch_size,
                           rnn_step=rnn_step, train_steps=int(train_steps), actor_gnn_layers=actor_gnn_layers,
                           Vl_gnn_layers=Vl_gnn_layers, cost_weight=cost_weight, cost_schedule=cost_schedule,
                           use_rnn=use_rnn, rnn_layers=rnn_layers, use_lstm=bool(use_lstm and use_rnn))
        self.device = env.device
        self.engine = EN.Engine(env.cfg, self.hp, self.device, T=env.max_episode_steps, allreduce=allreduce, world=world,
                                use_graphs=True, multi_stream=True, algo="informarl", rank=rank)
        self._init_dp(seed, world, rank)
        nc, lstm = _n_cells(use_rnn, rnn_layers), self.hp.use_lstm
        self.engine.policy.load_tree(INIT.init_policy(seed, node_dim, action_dim, actor_gnn_layers, nc, lstm))
        self.engine.Vl.load_tree(INIT.init_value(seed, node_dim, 1, Vl_gnn_layers, 2, rnn_layers=nc, lstm=lstm))
        self.engine.set_entropy_noise(int(np.random.randint(0, 102400)))
        self.init_rnn_state = torch.zeros(rnn_layers, n_agents, 2 if self.hp.use_lstm else 1, nets.HID, device=self.device)
        self._rng = np.random.default_rng([seed, 99])
        self._single = nets.Arena(self.device)

    @property
    def config(self) -> dict:          # informarl.py:200-221
        hp = self.hp
        return {"cost_weight": hp.cost_weight, "actor_gnn_layers": hp.actor_gnn_layers, "Vl_gnn_layers": hp.Vl_gnn_layers,
                "gamma": hp.gamma, "lr_actor": hp.lr_actor, "lr_Vl": hp.lr_Vl, "batch_size": hp.batch_size,
                "epoch_ppo": self.epoch_ppo, "clip_eps": hp.clip_eps, "gae_lambda": hp.gae_lambda, "coef_ent": hp.coef_ent,
                "max_grad_norm": hp.max_grad_norm, "seed": self.seed, "use_rnn": self.use_rnn, "rnn_layers": self.rnn_layers,
                "rnn_step": hp.rnn_step, "use_lstm": self.use_lstm, "cost_schedule": hp.cost_schedule}

    @property
    def params(self):
        e = self.engine
        return {"policy": e.policy.to_tree(), "Vl": e.Vl.to_tree()}

    def collect(self, params, keys):
        """one stochastic rollout per key (informarl.py:254-256); no deterministic companion rollout"""
        self._maybe_load(params)
        ro = self.engine.rollout(self._seeds(keys), True, noise_seed=int(self._rng.integers(1, 2 ** 62)))
        return self._wrap(ro)

    def update(self, rollout, step: int) -> dict:
        ro = self._last_rollouts.pop(id(rollout.actions), None)
        if ro is None:
            raise ValueError("update() needs a Rollout produced by this algo's collect()")
        self._last_rollouts.clear()
        info = {}
        for _ in range(self.epoch_ppo):                      # informarl.py:267-278
            info = self.engine.update(ro, None, int(step), self._perm(ro.B))   # host np.random on one device (informarl.py:270-271)
        return info

    # checkpoints: {dir}/{step}/{actor,Vl}.pkl (informarl.py:459-470)
    def save(self, save_dir: str, step: int):
        model_dir = os.path.join(save_dir, str(step))
        os.makedirs(model_dir, exist_ok=True)
        p = self.params
        for fname, key in (("actor.pkl", "policy"), ("Vl.pkl", "Vl")):
            with open(os.path.join(model_dir, fname), "wb") as f:
                CK.save_tree(p[key], f)

    def load(self, load_dir: str, step: int):
        path = os.path.join(load_dir, str(step))
        for fname, key in (("actor.pkl", "policy"), ("Vl.pkl", "Vl")):
            with open(os.path.join(path, fname), "rb") as f:   # weights-only unpickler: nothing in the file is executed
                self.engine.nets[key].load_tree(CK.load_tree(f))


class HCBFCRPO(InforMARL):
    """DGPPO with a hand-crafted CBF (dgppo/algo/hcbfcrpo.py:21-205): Vh := env.get_cost(graph); the actor and Vl as in
    DGPPO, no constraint-value network."""

    def __init__(self, env, node_dim: int, edge_dim: int, state_dim: int, action_dim: int, n_agents: int,
                 actor_gnn_layers: int = 2, Vl_gnn_layers: int = 2, Vh_gnn_layers: int = 1, gamma: float = 0.99,
                 lr_actor: float = 3e-4, lr_Vl: float = 1e-3, lr_Vh: float = 1e-3, batch_size: int = 8192,
                 epoch_ppo: int = 1, clip_eps: float = 0.25, gae_lambda: float = 0.95, coef_ent: float = 1e-2,
                 max_grad_norm: float = 2.0, seed: int = 0, use_rnn: bool = True, rnn_layers: int = 1, rnn_step: int = 16,
                 use_lstm: bool = False, alpha: float = 10.0, cbf_eps: float = 1e-2, cbf_weight: float = 1.0,
                 train_steps: int = 1e5, cbf_schedule: bool = True, allreduce=None, world: int = 1, rank: int = 0, **kwargs):
        Algorithm.__init__(self, env, node_dim, edge_dim, action_dim, n_agents)
        _check_rnn_options(use_rnn, use_lstm, rnn_layers)
        assert epoch_ppo >= 1
        assert node_dim == env.node_dim and action_dim == 2
        self.state_dim, self.seed = state_dim, seed
        self.epoch_ppo, self.use_rnn, self.rnn_layers, self.use_lstm = epoch_ppo, use_rnn, rnn_layers, use_lstm
        self.hp = EN.Hyper(gamma=gamma, gae_lambda=gae_lambda, clip_eps=clip_eps, coef_ent=coef_ent,
                           max_grad_norm=max_grad_norm, lr_actor=lr_actor, lr_Vl=lr_Vl, lr_Vh=lr_Vh, batch_size=batch_size,
                           rnn_step=rnn_step, alpha=alpha, cbf_eps=cbf_eps, cbf_weight=cbf_weight, cbf_schedule=cbf_schedule,
                           train_steps=int(train_steps), actor_gnn_layers=actor_gnn_layers, Vl_gnn_layers=Vl_gnn_layers,
                           Vh_gnn_layers=Vh_gnn_layers, use_rnn=use_rnn, rnn_layers=rnn_layers, use_lstm=bool(use_lstm and use_rnn))
        self.device = env.device
        self.engine = EN.Engine(env.cfg, self.hp, self.device, T=env.max_episode_steps, allreduce=allreduce, world=world,
                                use_graphs=True, multi_stream=True, algo="hcbfcrpo", rank=rank)
        self._init_dp(seed, world, rank)
        nc, lstm = _n_cells(use_rnn, rnn_layers), self.hp.use_lstm
        self.engine.policy.load_tree(INIT.init_policy(seed, node_dim, action_dim, actor_gnn_layers, nc, lstm))
        self.engine.Vl.load_tree(INIT.init_value(seed, node_dim, 1, Vl_gnn_layers, 2, rnn_layers=nc, lstm=lstm))
        self.engine.set_entropy_noise(int(np.random.randint(0, 102400)))
        self.init_rnn_state = torch.zeros(rnn_layers, n_agents, 2 if self.hp.use_lstm else 1, nets.HID, device=self.device)
        self._rng = np.random.default_rng([seed, 99])
        self._single = nets.Arena(self.device)

    @property
    def config(self) -> dict:          # the DGPPO config keys (hcbfcrpo.py inherits dgppo.py's property)
        return DGPPO.config.fget(self)


class InforMARLLagr(InforMARL):
    """InforMARL with a learned constraint value and Lagrange multipliers (dgppo/algo/informarl_lagr.py:25-327): the
    constraint-value net is DecRStateFn(use_global_info=True) with its own zero-initialised recurrent carry, trained on
    the stochastic rollout in chunks of rnn_step; advantage = -Al - mean_h(Ah * lambda[a,h]); after every policy step the
    multipliers move by lr_lagr along mean(Vh (1 - gamma) + rho Ah) and are clipped at zero."""

    def __init__(self, env, node_dim: int, edge_dim: int, state_dim: int, action_dim: int, n_agents: int,
                 actor_gnn_layers: int = 2, Vl_gnn_layers: int = 2, Vh_gnn_layers: int = 1, gamma: float = 0.99,
                 lr_actor: float = 3e-4, lr_Vl: float = 1e-3, lr_Vh: float = 1e-3, batch_size: int = 8192,
                 epoch_ppo: int = 1, clip_eps: float = 0.25, gae_lambda: float = 0.95, coef_ent: float = 1e-2,
                 max_grad_norm: float = 2.0, seed: int = 0, use_rnn: bool = True, rnn_layers: int = 1, rnn_step: int = 16,
                 use_lstm: bool = False, lagr_init: float = 0.78, lr_lagr: float = 1e-7, train_steps: int = 1e5,
                 allreduce=None, world: int = 1, rank: int = 0, **kwargs):
        Algorithm.__init__(self, env, node_dim, edge_dim, action_dim, n_agents)
        _check_rnn_options(use_rnn, use_lstm, rnn_layers)
        assert epoch_ppo >= 1
        assert node_dim == env.node_dim and action_dim == 2
        self.state_dim, self.seed = state_dim, seed
        self.epoch_ppo, self.use_rnn, self.rnn_layers, self.use_lstm = epoch_ppo, use_rnn, rnn_layers, use_lstm
        self.hp = EN.Hyper(gamma=gamma, gae_lambda=gae_lambda, clip_eps=clip_eps, coef_ent=coef_ent,
                           max_grad_norm=max_grad_norm, lr_actor=lr_actor, lr_Vl=lr_Vl, lr_Vh=lr_Vh, batch_size=batch_size,
                           rnn_step=rnn_step, train_steps=int(train_steps), actor_gnn_layers=actor_gnn_layers,
                           Vl_gnn_layers=Vl_gnn_layers, Vh_gnn_layers=Vh_gnn_layers, lagr_init=lagr_init, lr_lagr=lr_lagr,
                           use_rnn=use_rnn, rnn_layers=rnn_layers, use_lstm=bool(use_lstm and use_rnn))
        self.device = env.device
        self.engine = EN.Engine(env.cfg, self.hp, self.device, T=env.max_episode_steps, allreduce=allreduce, world=world,
                                use_graphs=True, multi_stream=True, algo="informarl_lagr", rank=rank)
        self._init_dp(seed, world, rank)
        nc, lstm = _n_cells(use_rnn, rnn_layers), self.hp.use_lstm
        self.engine.policy.load_tree(INIT.init_policy(seed, node_dim, action_dim, actor_gnn_layers, nc, lstm))
        self.engine.Vl.load_tree(INIT.init_value(seed, node_dim, 1, Vl_gnn_layers, 2, rnn_layers=nc, lstm=lstm))
        self.engine.Vh.load_tree(INIT.init_value(seed, node_dim, env.n_cost, Vh_gnn_layers, 3, global_info=True, rnn_layers=nc,
                                                 lstm=lstm))
        self.engine.set_entropy_noise(int(np.random.randint(0, 102400)))
        self.init_rnn_state = torch.zeros(rnn_layers, n_agents, 2 if self.hp.use_lstm else 1, nets.HID, device=self.device)
        self.init_Vh_rnn_state = torch.zeros(rnn_layers, n_agents, 1, nets.HID, device=self.device)
        self._rng = np.random.default_rng([seed, 99])
        self._single = nets.Arena(self.device)

    @property
    def ah_lagr(self):
        return self.engine.lagr

    @property
    def config(self) -> dict:          # informarl_lagr.py:109-116
        hp = self.hp
        return dict(InforMARL.config.fget(self), lr_Vh=hp.lr_Vh, Vh_gnn_layers=hp.Vh_gnn_layers, lagr_init=hp.lagr_init,
                    lr_lagr=hp.lr_lagr)

    @property
    def params(self):
        e = self.engine
        return {"policy": e.policy.to_tree(), "Vl": e.Vl.to_tree(), "Vh": e.Vh.to_tree()}

    # checkpoints: {dir}/{step}/{actor,Vl,Vh}.pkl (informarl_lagr.py:311-327)
    def save(self, save_dir: str, step: int):
        DGPPO.save(self, save_dir, step)

    def load(self, load_dir: str, step: int):
        DGPPO.load(self, load_dir, step)
