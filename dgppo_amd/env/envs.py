"""The five environment families on the hot path with the reference's PARAMS and state containers
(dgppo/env/lidar_env/{base,lidar_spread,lidar_target,lidar_bicycle_target}.py, dgppo/env/mpe/{base,mpe_spread,mpe_target}.py,
dgppo/env/obstacle.py:30-36)."""
from __future__ import annotations

from typing import NamedTuple, Optional

import torch

from .base import BatchState, MultiAgentEnv
from ..utils.graph import GraphsTuple


class Rectangle(NamedTuple):
    type: torch.Tensor
    center: torch.Tensor
    width: torch.Tensor
    height: torch.Tensor
    theta: torch.Tensor
    points: torch.Tensor


class LidarEnvState(NamedTuple):
    agent: torch.Tensor
    goal: torch.Tensor
    obstacle: Optional[Rectangle]

    @property
    def n_agent(self) -> int:
        return self.agent.shape[-2]


class MPEEnvState(NamedTuple):
    agent: torch.Tensor
    goal: torch.Tensor
    obs: Optional[torch.Tensor]

    @property
    def n_agent(self) -> int:
        return self.agent.shape[-2]


def _rect_from_records(rec: torch.Tensor) -> Rectangle:
    """[.., n_obs, 16] records -> Rectangle fields (obstacle.py:30-36)."""
    return Rectangle(torch.zeros(rec.shape[:-1] + (1,), device=rec.device), rec[..., 0:2], rec[..., 2], rec[..., 3],
                     rec[..., 4], rec[..., 8:16].reshape(rec.shape[:-1] + (4, 2)))


def _records_from_rect(r: Rectangle) -> torch.Tensor:
    rec = torch.zeros(r.center.shape[:-1] + (16,), device=r.center.device)
    rec[..., 0:2] = r.center
    rec[..., 2], rec[..., 3], rec[..., 4] = r.width, r.height, r.theta
    rec[..., 5], rec[..., 6] = torch.cos(r.theta), torch.sin(r.theta)
    rec[..., 8:16] = r.points.reshape(r.points.shape[:-2] + (8,))
    return rec


class _LidarEnv(MultiAgentEnv):
    AGENT, GOAL, OBS = 0, 1, 2
    PARAMS = {"car_radius": 0.05, "comm_radius": 0.5, "n_rays": 32, "obs_len_range": [0.1, 0.3], "n_obs": 3,
              "default_area_size": 1.5, "dist2goal": 0.01, "top_k_rays": 8}

    def _env_states(self, st: BatchState):
        return LidarEnvState(st.agent, st.goal, _rect_from_records(st.obst) if st.obst is not None else None)

    def _state_of(self, graph: GraphsTuple) -> BatchState:
        n, k = self.num_agents, self.cfg.top_k
        es = graph.env_states
        states = graph.states
        ng = self.cfg.n_goals
        agent = states[..., :n, :].reshape(1, n, self.state_dim).contiguous()
        goal = states[..., n:n + ng, :].reshape(1, ng, self.state_dim).contiguous()
        obst = hits = None
        if self.cfg.n_obs > 0:
            obst = _records_from_rect(es.obstacle).reshape(1, self.cfg.n_obs, 16).contiguous()
            hits = states[..., n + ng:n + ng + n * k, :2].reshape(1, n, k, 2).contiguous()
        return BatchState(agent, goal, obst, hits)


    def _batch_of_env_state(self, env_state: LidarEnvState, lidar_data) -> BatchState:
        n, ng, sd, dev = self.num_agents, self.cfg.n_goals, self.state_dim, self.device
        f = lambda x, *shape: torch.as_tensor(x, dtype=torch.float32, device=dev).reshape(*shape).contiguous()
        obst = hits = None
        if self.cfg.n_obs > 0:
            obst = _records_from_rect(Rectangle(*[torch.as_tensor(v, dtype=torch.float32, device=dev)
                                                  for v in env_state.obstacle])).reshape(1, self.cfg.n_obs, 16).contiguous()
            if lidar_data is not None:
                hits = f(lidar_data, 1, n, self.cfg.top_k, 2)
        return BatchState(f(env_state.agent, 1, n, sd), f(env_state.goal, 1, ng, sd), obst, hits)


class LidarSpread(_LidarEnv):
    KIND = "LidarSpread"
    PARAMS = dict(_LidarEnv.PARAMS)


class LidarTarget(_LidarEnv):
    KIND = "LidarTarget"
    PARAMS = dict(_LidarEnv.PARAMS)


class LidarBicycleTarget(_LidarEnv):
    KIND = "LidarBicycleTarget"
    PARAMS = dict(_LidarEnv.PARAMS)


class _MPE(MultiAgentEnv):
    AGENT, GOAL, OBS = 0, 1, 2
    PARAMS = {"car_radius": 0.05, "comm_radius": 0.5, "n_obs": 3, "obs_radius": 0.05, "default_area_size": 1.5,
              "dist2goal": 0.01}

    def _env_states(self, st: BatchState):
        return MPEEnvState(st.agent, st.goal, st.obst)

    def _state_of(self, graph: GraphsTuple) -> BatchState:
        n = self.num_agents
        states = graph.states
        ng = self.cfg.n_goals
        agent = states[..., :n, :].reshape(1, n, 4).contiguous()
        goal = states[..., n:n + ng, :].reshape(1, ng, 4).contiguous()
        obst = states[..., n + ng:n + ng + self.cfg.n_obs, :].reshape(1, self.cfg.n_obs, 4).contiguous() if self.cfg.n_obs > 0 else None
        return BatchState(agent, goal, obst, None)


    def _batch_of_env_state(self, env_state: MPEEnvState, lidar_data=None) -> BatchState:
        n, ng, dev = self.num_agents, self.cfg.n_goals, self.device
        f = lambda x, *shape: torch.as_tensor(x, dtype=torch.float32, device=dev).reshape(*shape).contiguous()
        obst = f(env_state.obs, 1, self.cfg.n_obs, 4) if self.cfg.n_obs > 0 else None
        return BatchState(f(env_state.agent, 1, n, 4), f(env_state.goal, 1, ng, 4), obst, None)


class MPESpread(_MPE):
    KIND = "MPESpread"
    PARAMS = dict(_MPE.PARAMS)


# ---- task variants: same kernels; reset, reward goals and goal-node count differ ----------------------
class LidarLine(_LidarEnv):
    """dgppo/env/lidar_env/lidar_line.py: two landmark nodes; the n reward goals divide the segment between them."""
    KIND = "LidarLine"
    PARAMS = dict(_LidarEnv.PARAMS)

    def landmark2goal(self, landmarks):
        n = self.num_agents
        return landmarks[..., 0:1, :] + torch.arange(n, device=landmarks.device, dtype=torch.float32)[:, None] * \
            (landmarks[..., 1:2, :] - landmarks[..., 0:1, :]) / (n - 1)


class MPELine(_MPE):
    """dgppo/env/mpe/mpe_line.py (n <= 3: the goals are the interior division points, :126-128)."""
    KIND = "MPELine"
    PARAMS = dict(_MPE.PARAMS)

    def landmark2goal(self, landmarks):
        n = self.num_agents
        idx, den = (torch.arange(1, n + 1), n + 1) if n <= 3 else (torch.arange(0, n), n - 1)
        idx = idx.to(device=landmarks.device, dtype=torch.float32)
        return landmarks[..., 0:1, :] + idx[:, None] * (landmarks[..., 1:2, :] - landmarks[..., 0:1, :]) / den


class MPEFormation(_MPE):
    """dgppo/env/mpe/mpe_formation.py: one landmark node; the reward goals lie on a circle of radius comm_radius."""
    KIND = "MPEFormation"
    PARAMS = dict(_MPE.PARAMS)


class MPECorridor(_MPE):
    """dgppo/env/mpe/mpe_corridor.py: two fixed discs leave a corridor; agents start below it, goals lie above."""
    KIND = "MPECorridor"
    PARAMS = {"car_radius": 0.05, "comm_radius": 0.5, "default_area_size": 1.0, "dist2goal": 0.01, "n_obs": 2,
              "corridor_width": 0.2}


class MPEConnectSpread(_MPE):
    """dgppo/env/mpe/mpe_connect_spread.py: a third cost keeps the team connected while it passes one large disc."""
    KIND = "MPEConnectSpread"
    PARAMS = {"car_radius": 0.05, "comm_radius": 0.5, "default_area_size": 1.0, "dist2goal": 0.01, "n_obs": 1,
              "obs_radius": 0.25, "connect_radius": 0.45}


class MPETarget(_MPE):
    """n_obs == 0 is guarded like MPESpread; the unmodified reference raises TypeError there."""
    KIND = "MPETarget"
    PARAMS = dict(_MPE.PARAMS)
