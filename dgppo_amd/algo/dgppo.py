"""DGPPO with the reference's constructor / method surface (dgppo/algo/dgppo.py:27-321 through
informarl_lagr.py:27-123 and informarl.py:30-256), driving the HIP engine.

Differences that cannot be avoided and are recorded in DESIGN.md: PRNG keys are integer seeds (JAX threefry streams are not
reproducible); `params` are flax-named trees of numpy arrays snapshotted from the device buffers."""
from __future__ import annotations

import os
from typing import Optional

import numpy as np
import torch

from .. import engine as EN
from .. import init as INIT
from .. import nets
from .. import ops_nn as K
from ..trainer.data import Rollout
from ..utils import checkpoint as CK
from ..utils.graph import GraphsTuple
from .base import Algorithm


class _LazyGraphs:
    """rollout.graph / rollout.next_graph: GraphsTuple fields of all (env, t), materialised on first access."""

    def __init__(self, env, ro: EN.RolloutData, offset: int):
        self._env, self._ro, self._off, self._g = env, ro, offset, None

    def _mat(self):
        if self._g is None:
            ro, T, off = self._ro.finalize(), self._ro.T, self._off
            B = ro.B
            flat = lambda x: None if x is None else x[:, off:off + T].reshape((B * T,) + x.shape[2:]).contiguous()
            rep = lambda x: None if x is None else x.repeat_interleave(T, dim=0)
            from ..env.base import BatchState
            g = self._env.graph_batch(BatchState(flat(ro.agent), rep(ro.goal), rep(ro.obst), flat(ro.hits)))
            unf = lambda x: x.view((B, T) + x.shape[1:]) if torch.is_tensor(x) else x
            es = g.env_states
            es = type(es)(*[_unflatten_tree(v, B, T) for v in es])
            self._g = GraphsTuple(*[unf(getattr(g, f)) for f in GraphsTuple._fields[:8]], es, None)
        return self._g

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self._mat(), name)

    def _replace(self, **kw):
        return self._mat()._replace(**kw)


def _unflatten_tree(v, B, T):
    if torch.is_tensor(v):
        return v.view((B, T) + v.shape[1:])
    if isinstance(v, tuple) and hasattr(v, "_fields"):
        return type(v)(*[_unflatten_tree(x, B, T) for x in v])
    return v


def _check_rnn_options(use_rnn: bool, use_lstm: bool, rnn_layers: int):
    """GRU or LSTM cells, any number of stacked layers (dgppo/nn/rnn.py:17-29), or --no-rnn"""
    assert rnn_layers >= 1


def _n_cells(use_rnn: bool, rnn_layers: int) -> int:
    return rnn_layers if use_rnn else 0


class DGPPO(Algorithm):
    def __init__(self, env, node_dim: int, edge_dim: int, state_dim: int, action_dim: int, n_agents: int,
                 actor_gnn_layers: int = 2, Vl_gnn_layers: int = 2, Vh_gnn_layers: int = 1, gamma: float = 0.99,
                 lr_actor: float = 3e-4, lr_Vl: float = 1e-3, lr_Vh: float = 1e-3, batch_size: int = 8192,
                 epoch_ppo: int = 1, clip_eps: float = 0.25, gae_lambda: float = 0.95, coef_ent: float = 1e-2,
                 max_grad_norm: float = 2.0, seed: int = 0, use_rnn: bool = True, rnn_layers: int = 1, rnn_step: int = 16,
                 use_lstm: bool = False, alpha: float = 10.0, cbf_eps: float = 1e-2, cbf_weight: float = 1.0,
                 train_steps: int = 1e5, cbf_schedule: bool = True, allreduce=None, world: int = 1, rank: int = 0, **kwargs):
        super().__init__(env, node_dim, edge_dim, action_dim, n_agents)
        _check_rnn_options(use_rnn, use_lstm, rnn_layers)
        assert epoch_ppo >= 1
        assert node_dim == env.node_dim and action_dim == 2
        self.state_dim = state_dim
        self.seed = seed
        self.epoch_ppo, self.use_rnn, self.rnn_layers, self.use_lstm = epoch_ppo, use_rnn, rnn_layers, use_lstm
        self.hp = EN.Hyper(gamma=gamma, gae_lambda=gae_lambda, clip_eps=clip_eps, coef_ent=coef_ent,
                           max_grad_norm=max_grad_norm, lr_actor=lr_actor, lr_Vl=lr_Vl, lr_Vh=lr_Vh, batch_size=batch_size,
                           rnn_step=rnn_step, alpha=alpha, cbf_eps=cbf_eps, cbf_weight=cbf_weight, cbf_schedule=cbf_schedule,
                           train_steps=int(train_steps), actor_gnn_layers=actor_gnn_layers, Vl_gnn_layers=Vl_gnn_layers,
                           Vh_gnn_layers=Vh_gnn_layers, use_rnn=use_rnn, rnn_layers=rnn_layers, use_lstm=bool(use_lstm and use_rnn))
        self.device = env.device
        self.engine = EN.Engine(env.cfg, self.hp, self.device, T=env.max_episode_steps, allreduce=allreduce, world=world,
                                use_graphs=True, multi_stream=True, rank=rank)
        self._init_dp(seed, world, rank)
        nc, lstm = _n_cells(use_rnn, rnn_layers), self.hp.use_lstm
        self.engine.policy.load_tree(INIT.init_policy(seed, node_dim, action_dim, actor_gnn_layers, nc, lstm))
        self.engine.Vl.load_tree(INIT.init_value(seed, node_dim, 1, Vl_gnn_layers, 2, rnn_layers=nc, lstm=lstm))
        # the constraint-value net is built with ValueNet's default of one cell (dgppo.py:83-95)
        self.engine.Vh.load_tree(INIT.init_value(seed, node_dim, env.n_cost, Vh_gnn_layers, 3, rnn_layers=min(nc, 1)))
        # np.random.randint(0, 102400) at trace time in the reference (distribution.py:40)
        self.engine.set_entropy_noise(int(np.random.randint(0, 102400)))
        # (n_rnn_layers, n_agents, n_carries, rnn_state_dim), zeros (informarl.py:115-124)
        # (n_rnn_layers, n_agents, n_carries, 64): one carry for GRU / no cell, (c, h) for LSTM (informarl.py:115-124)
        self.init_rnn_state = torch.zeros(rnn_layers, n_agents, 2 if self.hp.use_lstm else 1, nets.HID, device=self.device)
        self._rng = np.random.default_rng([seed, 99])
        self._single = nets.Arena(self.device)

    # ---- data parallelism: batch_size and the keys handed to collect() are this rank's SHARE; everything random
    # that must agree across the ranks comes from generators every rank seeds identically ----
    def _init_dp(self, seed: int, world: int, rank: int):
        self.world, self.rank = int(world), int(rank)
        self._perm_rng = np.random.default_rng([seed, 4242])
        self.perm_fn = None                   # tests: B -> permutation of this rank's env indices

    def _perm(self, B: int) -> np.ndarray:
        """minibatch order of one epoch.  One device: host np.random like the reference (dgppo.py:155-156).  Several ranks: the
        SAME permutation of the local env indices on every rank, from a generator all ranks seed identically — minibatch k
        of the job is the union of the ranks' k-th slices."""
        if self.perm_fn is not None:
            return np.asarray(self.perm_fn(B))
        if self.world == 1:
            perm = np.arange(B)
            np.random.shuffle(perm)
            return perm
        return self._perm_rng.permutation(B)

    def _draw_keys(self, B_local: int) -> np.ndarray:
        """B_local fresh keys of THIS rank out of world * B_local drawn identically on every rank (a function of the global
        env index, so the union over the ranks is what one device would draw for the global batch)"""
        x = self._rng.integers(1, 2 ** 62, size=B_local * self.world)
        return x[self.rank * B_local:(self.rank + 1) * B_local]

    # ---- carry layout: the reference's (n_layers, n_agents, n_carries = 1, 64) <-> the engine's packed rows [n, L * 64] ----
    def _carry_dims(self):
        nc = 2 if self.hp.use_lstm else 1
        return max(self.engine.HC // (nets.HID * nc), 1), nc

    def _pack_carry(self, rnn_state) -> torch.Tensor:
        """(L, n, n_carries, 64) -> packed rows [n, L * n_carries * 64] (per layer: carry 0, carry 1)"""
        n, (L, nc) = self.n_agents, self._carry_dims()
        x = torch.as_tensor(rnn_state, dtype=torch.float32, device=self.device).reshape(-1, n, nc, nets.HID)[:L]
        return x.permute(1, 0, 2, 3).reshape(n, self.engine.HC).contiguous()

    def _unpack_carry(self, rows: torch.Tensor) -> torch.Tensor:
        """[..., n, L * n_carries * 64] -> [..., L, n, n_carries, 64]"""
        n, (L, nc) = self.n_agents, self._carry_dims()
        lead = rows.shape[:-2]
        x = rows.reshape(lead + (n, L, nc, nets.HID))
        k = len(lead)
        return x.permute(tuple(range(k)) + (k + 1, k, k + 2, k + 3))

    # ---- reference properties ----
    @property
    def config(self) -> dict:
        hp = self.hp
        return {"cost_weight": 0.0, "actor_gnn_layers": hp.actor_gnn_layers, "Vl_gnn_layers": hp.Vl_gnn_layers,
                "gamma": hp.gamma, "lr_actor": hp.lr_actor, "lr_Vl": hp.lr_Vl, "batch_size": hp.batch_size,
                "epoch_ppo": self.epoch_ppo, "clip_eps": hp.clip_eps, "gae_lambda": hp.gae_lambda, "coef_ent": hp.coef_ent,
                "max_grad_norm": hp.max_grad_norm, "seed": self.seed, "use_rnn": self.use_rnn, "rnn_layers": self.rnn_layers,
                "rnn_step": hp.rnn_step, "use_lstm": self.use_lstm, "cost_schedule": False, "lr_Vh": hp.lr_Vh,
                "Vh_gnn_layers": hp.Vh_gnn_layers, "lagr_init": 0.78, "lr_lagr": 1e-7, "alpha": hp.alpha,
                "cbf_eps": hp.cbf_eps, "cbf_weight": hp.cbf_weight, "cbf_schedule": hp.cbf_schedule}

    @property
    def params(self):
        e = self.engine
        return {"policy": e.policy.to_tree(), "Vl": e.Vl.to_tree(), "Vh": e.Vh.to_tree()}

    def _maybe_load(self, params):
        if params is not None:
            for k, net in self.engine.nets.items():
                if k in params:
                    net.load_tree(params[k])

    # ---- single-graph act / step (B = 1 view of the batched kernels) ----
    def _policy_single(self, graph: GraphsTuple, rnn_state):
        env, cfg = self._env, self._env.cfg
        st = env._state_of(graph)
        n = cfg.n_agents
        feats = self.engine._feats_at("one", st.agent, st.hits, st.goal, st.obst, 1)
        h0 = self._pack_carry(rnn_state)
        hs = torch.empty(n, self.engine.HC, device=self.device)
        act = self.engine.policy.forward(feats, n_seq=n, T=1, h0=h0, tag="one", hs_out=hs, train=False)
        return act["ms"], self._unpack_carry(hs)

    def act(self, graph: GraphsTuple, rnn_state, params=None):
        self._maybe_load(params)
        ms, new_rnn = self._policy_single(graph, rnn_state)
        action = torch.empty(self.n_agents, 2, device=self.device)
        K.policy_head(ms, None, None, action, None, None, self.n_agents, 1)
        return action, new_rnn

    def step(self, graph: GraphsTuple, rnn_state, key, params=None):
        self._maybe_load(params)
        ms, new_rnn = self._policy_single(graph, rnn_state)
        from .. import ops_env as OE
        eps = torch.empty(self.n_agents, 2, device=self.device)
        OE.randn(int(key), 0, eps.view(-1))
        action = torch.empty(self.n_agents, 2, device=self.device)
        log_pi = torch.empty(self.n_agents, device=self.device)
        K.policy_head(ms, eps, None, action, log_pi, None, self.n_agents, 0)
        return action, log_pi, new_rnn

    # ---- collect / update ----
    def _seeds(self, keys) -> torch.Tensor:
        if torch.is_tensor(keys):
            return keys.to(self.device, torch.int64)
        return torch.from_numpy(np.ascontiguousarray(np.asarray(keys).astype(np.uint64).view(np.int64))).to(self.device)

    def _wrap(self, ro: EN.RolloutData, env=None) -> Rollout:
        ro.finalize()
        env = self._env if env is None else env
        r = Rollout(_LazyGraphs(env, ro, 0), ro.actions, self._unpack_carry(ro.rnn_states), ro.rewards, ro.costs,
                    torch.zeros(ro.B, ro.T, dtype=torch.bool, device=self.device), ro.log_pis, _LazyGraphs(env, ro, 1))
        self._last_rollouts = getattr(self, "_last_rollouts", {})
        self._last_rollouts[id(r.actions)] = ro
        return r

    def collect(self, params, keys) -> Rollout:
        """algo.collect(params, keys): one stochastic rollout per key (informarl.py:254-256)."""
        self._maybe_load(params)
        seeds = self._seeds(keys)
        # the deterministic rollout that update() needs for the constraint-value targets (dgppo.py:139-141) uses the same
        # parameters as this collect: it is launched alongside on a second stream and handed to update()
        det_seeds = self._seeds(self._draw_keys(int(seeds.shape[0])))
        ro, det = self.engine.rollout_pair(seeds, det_seeds, noise_seed=int(self._rng.integers(1, 2 ** 62)))
        self._pending_det = (ro, det)
        return self._wrap(ro)

    def collect_deterministic(self, keys, env=None) -> Rollout:
        eng = self.engine
        if env is not None and env is not self._env:
            assert env.cfg.kind == self._env.cfg.kind and env.num_agents == self.n_agents
        seeds = self._seeds(keys)
        pend = getattr(self, "_pending_det", None)
        if pend is not None and pend[1].B == int(seeds.shape[0]):
            # this rollout reuses the record buffers of the deterministic rollout that collect() prepared for update():
            # drop the hand-over, update() will produce its own
            self._pending_det = None
        ro = eng.rollout(seeds, False)
        return self._wrap(ro, env)

    def collect_stochastic(self, keys, env=None) -> Rollout:
        """stochastic test rollouts (test.py:78-83 with --stochastic: algo.step inside test_rollout), batched"""
        if env is not None and env is not self._env:
            assert env.cfg.kind == self._env.cfg.kind and env.num_agents == self.n_agents
        ro = self.engine.rollout(self._seeds(keys), True, noise_seed=int(self._rng.integers(1, 2 ** 62)))
        return self._wrap(ro, env)

    def update(self, rollout: Rollout, step: int) -> dict:
        ro = self._last_rollouts.pop(id(rollout.actions), None)
        if ro is None:
            raise ValueError("update() needs a Rollout produced by this algo's collect()")
        self._last_rollouts.clear()
        # deterministic rollout for the constraint-value targets (dgppo.py:139-141): produced next to collect()
        pend = getattr(self, "_pending_det", None)
        self._pending_det = None
        if pend is not None and pend[0] is ro:
            det = pend[1]
        else:
            det = self.engine.rollout(self._seeds(self._draw_keys(ro.B)), False)
        info = {}
        for _ in range(self.epoch_ppo):                      # dgppo.py:154-172: every epoch reshuffles and recomputes the targets
            # with the current parameters; the det rollout is shared (:140-141)
            info = self.engine.update(ro, det, int(step), self._perm(ro.B))
        return info

    # ---- checkpoints: {dir}/{step}/{actor,Vl,Vh}.pkl with flax-named params (informarl_lagr.py:311-327) ----
    def save(self, save_dir: str, step: int):
        model_dir = os.path.join(save_dir, str(step))
        os.makedirs(model_dir, exist_ok=True)
        p = self.params
        for fname, key in (("actor.pkl", "policy"), ("Vl.pkl", "Vl"), ("Vh.pkl", "Vh")):
            with open(os.path.join(model_dir, fname), "wb") as f:
                CK.save_tree(p[key], f)

    def load(self, load_dir: str, step: int):
        path = os.path.join(load_dir, str(step))
        for fname, key in (("actor.pkl", "policy"), ("Vl.pkl", "Vl"), ("Vh.pkl", "Vh")):
            with open(os.path.join(path, fname), "rb") as f:   # weights-only unpickler: nothing in the file is executed
                self.engine.nets[key].load_tree(CK.load_tree(f))
