"""MultiAgentEnv with the reference's attribute / method surface (dgppo/env/base.py:30-150), backed by the batched HIP
kernels.  The reference's single-graph methods (`reset(key)`, `step(graph, action)`) are the B = 1 view of the batched
ones (`reset_batch`, `step_batch`) that the engine uses."""
from __future__ import annotations

from abc import ABC
from typing import NamedTuple, Optional, Tuple

import numpy as np
import torch

from .. import _native as N
from .. import ops_env as OE
from ..utils.graph import GraphsTuple


class StepResult(NamedTuple):
    graph: GraphsTuple
    reward: torch.Tensor
    cost: torch.Tensor
    done: torch.Tensor
    info: dict


class BatchState(NamedTuple):
    """compact batched env state (device tensors)."""
    agent: torch.Tensor            # [B, n, sd]
    goal: torch.Tensor             # [B, n, sd]
    obst: Optional[torch.Tensor]   # LiDAR: [B, n_obs, 16] rectangle records; MPE: [B, n_obs, sd]
    hits: Optional[torch.Tensor]   # LiDAR: [B, n, k, 2]


class MultiAgentEnv(ABC):
    PARAMS: dict = {}
    KIND: str = ""

    def __init__(self, num_agents: int, area_size: Optional[float] = None, max_step: int = 128, dt: float = 0.03,
                 params: Optional[dict] = None, device: Optional[torch.device] = None):
        self._params = dict(self.PARAMS) if params is None else params
        self._num_agents = num_agents
        self._area_size = self._params["default_area_size"] if area_size is None else area_size
        self._dt = dt
        self._max_step = max_step
        self._device = device
        p = self._params
        self.cfg = N.make_env_cfg(N.ENV_KINDS[self.KIND], num_agents, p["n_obs"], p.get("n_rays", 32), p.get("top_k_rays", 8),
                                  self._area_size, dt, p["car_radius"], p["comm_radius"], p.get("obs_radius"),
                                  p["dist2goal"], p.get("connect_radius", 0.45), p.get("corridor_width", 0.2))
        self.num_goals = self.cfg.n_goals                       # goal NODES: n, 2 landmarks (Line) or 1 (Formation)
        # what the kind fixes is written back, as the reference does (mpe_corridor.py:35-39, mpe_connect_spread.py:38-40)
        p["n_obs"] = self.cfg.n_obs
        if not self.cfg.is_lidar:
            p["obs_radius"] = float(np.float32(self.cfg.obs_radius)) if "obs_radius" not in p else p["obs_radius"]
        self._ray = None

    # ---- reference attribute surface ----
    @property
    def params(self) -> dict:
        return self._params

    @property
    def num_agents(self) -> int:
        return self._num_agents

    @property
    def area_size(self) -> float:
        return self._area_size

    @property
    def dt(self) -> float:
        return self._dt

    @property
    def max_episode_steps(self) -> int:
        return self._max_step

    @property
    def n_cost(self) -> int:
        return self.cfg.n_cost

    @property
    def cost_components(self) -> Tuple[str, ...]:
        base = ("agent collisions", "obs collisions")
        return base + ("connectivity",) if self.cfg.n_cost == 3 else base          # mpe_connect_spread.py:50-52

    @property
    def state_dim(self) -> int:
        return self.cfg.state_dim

    @property
    def node_dim(self) -> int:
        return self.cfg.node_dim

    @property
    def edge_dim(self) -> int:
        return 4

    @property
    def action_dim(self) -> int:
        return 2

    @property
    def device(self) -> torch.device:
        if self._device is None:
            if not torch.cuda.is_available():
                raise RuntimeError("dgppo_amd environments run on the GPU only (no CPU fallback)")
            self._device = torch.device("cuda", torch.cuda.current_device())
        return self._device

    def state_lim(self, state=None):
        a = self.area_size
        if self.cfg.kind == 2:
            return (torch.tensor([0., 0., -1., -1., -0.5]), torch.tensor([a, a, 1., 1., 0.5]))
        v = 0.5 if self.cfg.is_lidar else 1.0
        return torch.tensor([0., 0., -v, -v]), torch.tensor([a, float(self.cfg.y_limit), v, v])   # y: 2 a for Corridor / ConnectSpread

    def action_lim(self):
        return -torch.ones(2), torch.ones(2)

    def clip_state(self, state):
        lo, hi = self.state_lim()
        return torch.minimum(torch.maximum(state, lo.to(state.device)), hi.to(state.device))

    def clip_action(self, action):
        return torch.clamp(action, -1.0, 1.0)

    # ---- batched interface (what the engine uses) ----
    def _rays(self):
        if self._ray is None and self.cfg.is_lidar:
            self._ray = OE.ray_tables(self.cfg.n_rays, self.device)
        return self._ray if self._ray is not None else (None, None)

    @property
    def _has_hits(self):
        return self.cfg.is_lidar and self.cfg.n_obs > 0

    def reset_batch(self, seeds, want_graph: bool = False):
        """seeds: int64 tensor / array [B] -> BatchState (and the GraphsTuple batch if asked)."""
        cfg, dev = self.cfg, self.device
        seeds = torch.as_tensor(np.asarray(seeds, dtype=np.int64) if not torch.is_tensor(seeds) else seeds).to(dev)
        B = int(seeds.shape[0])
        n, sd = cfg.n_agents, cfg.state_dim
        agent = torch.empty(B, n, sd, device=dev)
        goal = torch.empty(B, cfg.n_goals, sd, device=dev)
        obst = torch.empty(B, cfg.n_obs, cfg.obst_stride, device=dev) if cfg.n_obs > 0 else None
        n_failed = torch.zeros(1, dtype=torch.int32, device=dev)
        OE.env_reset(cfg, seeds, agent, goal, obst, n_failed)
        if int(n_failed.item()):       # the API path may sync: an invalid scene never leaves reset()
            raise RuntimeError(f"env reset: {int(n_failed.item())} of {B} scenes could not be placed within the kernels' "
                               f"rejection-loop bounds (too many agents / obstacles for the area?)")
        hits, g = None, None
        rc, rs = self._rays()
        if self._has_hits or want_graph:
            hits = torch.empty(B, n, cfg.top_k, 2, device=dev) if self._has_hits else None
            g = OE.alloc_graph(cfg, B, dev) if want_graph else None
            if self._has_hits:
                OE.env_step(cfg, agent, None, goal, obst, None, rc, rs, None, hits, None, None, g)
            else:
                OE.graph_materialize(cfg, agent, goal, obst, None, g)
        st = BatchState(agent, goal, obst, hits)
        return (st, self._graphs(st, g)) if want_graph else st

    def step_batch(self, st: BatchState, action: torch.Tensor, want_graph: bool = False):
        """-> (next BatchState, reward [B], cost [B,n,2][, GraphsTuple batch])."""
        cfg, dev = self.cfg, self.device
        B = st.agent.shape[0]
        n = cfg.n_agents
        nx = torch.empty_like(st.agent)
        nh = torch.empty_like(st.hits) if st.hits is not None else None
        rew = torch.empty(B, device=dev)
        cost = torch.empty(B, n, cfg.n_cost, device=dev)
        g = OE.alloc_graph(cfg, B, dev) if want_graph else None
        rc, rs = self._rays()
        OE.env_step(cfg, st.agent, action.contiguous(), st.goal, st.obst, st.hits, rc, rs, nx, nh, rew, cost, g)
        nst = BatchState(nx, st.goal, st.obst, nh)
        if want_graph:
            return nst, rew, cost, self._graphs(nst, g)
        return nst, rew, cost

    def graph_batch(self, st: BatchState) -> GraphsTuple:
        g = OE.alloc_graph(self.cfg, st.agent.shape[0], self.device)
        OE.graph_materialize(self.cfg, st.agent, st.goal, st.obst, st.hits, g)
        return self._graphs(st, g)

    def _env_states(self, st: BatchState):
        raise NotImplementedError

    def _graphs(self, st: BatchState, g: dict) -> GraphsTuple:
        return GraphsTuple(g["n_node"], g["n_edge"], g["nodes"], g["edges"], g["states"], g["receivers"], g["senders"],
                           g["node_type"], self._env_states(st), None)

    # ---- reference single-graph interface = B = 1 view ----
    @staticmethod
    def _squeeze(graph: GraphsTuple) -> GraphsTuple:
        sq = lambda x: x[0] if torch.is_tensor(x) else x
        es = graph.env_states
        es = type(es)(*[_sq_tree(v) for v in es])
        return GraphsTuple(*[sq(getattr(graph, f)) for f in GraphsTuple._fields[:8]], es, None)

    def reset(self, key) -> GraphsTuple:
        """key: an integer seed (the reference passes a JAX PRNGKey; its threefry stream is not reproducible here)."""
        seed = int(key.reshape(-1)[-1]) if hasattr(key, "reshape") else int(key)
        _, g = self.reset_batch(np.array([seed], dtype=np.int64), want_graph=True)
        return self._squeeze(g)

    def _state_of(self, graph: GraphsTuple) -> BatchState:
        raise NotImplementedError

    def step(self, graph: GraphsTuple, action, get_eval_info: bool = False) -> StepResult:
        st = self._state_of(graph)
        action = torch.as_tensor(action, dtype=torch.float32, device=self.device).reshape(1, self.num_agents, 2)
        nst, rew, cost, g = self.step_batch(st, action, want_graph=True)
        return StepResult(self._squeeze(g), rew[0], cost[0], torch.tensor(False), {})

    def get_cost(self, graph: GraphsTuple):
        st = self._state_of(graph)
        zero = torch.zeros(1, self.num_agents, 2, device=self.device)
        return self.step_batch(st, zero)[2][0]                # cost is a function of the pre-step graph only

    def _batch_of_env_state(self, env_state, lidar_data) -> BatchState:
        raise NotImplementedError

    def get_graph(self, env_state, lidar_data=None) -> GraphsTuple:
        """GraphsTuple of ONE environment state (dgppo/env/lidar_env/base.py:227-271 `get_graph(state, lidar_data)`,
        dgppo/env/mpe/base.py:211-241 `get_graph(env_state)`): an adapter over the batched `graph_batch` (B = 1).  For a LiDAR
        env with obstacles `lidar_data` are the top-k hit points, [n, k, 2] (or merged [n * k, 2]) as `get_lidar_vmap`
        returns them; when omitted they are sensed from the state first."""
        st = self._batch_of_env_state(env_state, lidar_data)
        if self._has_hits and st.hits is None:
            hits = torch.empty(1, self.num_agents, self.cfg.top_k, 2, device=self.device)
            rc, rs = self._rays()
            OE.env_step(self.cfg, st.agent, None, st.goal, st.obst, None, rc, rs, None, hits, None, None, None)
            st = BatchState(st.agent, st.goal, st.obst, hits)
        return self._squeeze(self.graph_batch(st))

    def render_video(self, rollout, video_path, Ta_is_unsafe=None, viz_opts: Optional[dict] = None, dpi: int = 100, **kwargs):
        """One episode as an animation (dgppo/env/lidar_env/base.py:209-221, dgppo/env/mpe/base.py render_video).  `rollout`
        holds one episode ([T, ...]) or a batch ([B, T, ...], pass index=b).  Returns the path written (`.gif` when no
        ffmpeg binary is available for `.mp4`)."""
        from . import plot
        p = self._params
        common = dict(rollout=rollout, video_path=video_path, side_length=self.area_size, dim=2, n_agent=self.num_agents,
                      r=p["car_radius"], cost_components=self.cost_components, Ta_is_unsafe=Ta_is_unsafe, viz_opts=viz_opts,
                      n_goal=self.num_goals, dpi=dpi, **kwargs)
        if self.cfg.is_lidar:
            return plot.render_lidar(n_rays=self.cfg.top_k if self.cfg.n_obs > 0 else 0, **common)
        return plot.render_mpe(n_obs=self.cfg.n_obs, obs_r=p.get("obs_radius", 0.05), **common)


def _sq_tree(v):
    if torch.is_tensor(v):
        return v[0]
    if isinstance(v, tuple) and hasattr(v, "_fields"):
        return type(v)(*[_sq_tree(x) for x in v])
    return v
