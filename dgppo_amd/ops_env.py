"""Torch-tensor wrappers over the environment entry points of the C ABI (include/dgppo_hip.h).
Tensors are storage only; all arithmetic runs in the HIP kernels."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np
import torch

from . import _native as N


def ray_tables(n_rays: int, device) -> tuple[torch.Tensor, torch.Tensor]:
    """cos/sin(linspace(-pi, pi-2pi/R, R)) in fp32 (dgppo/env/utils.py:51)."""
    thetas = np.linspace(-np.pi, np.pi - 2 * np.pi / n_rays, n_rays).astype(np.float32)
    rc = torch.from_numpy(np.cos(thetas).astype(np.float32)).to(device)
    rs = torch.from_numpy(np.sin(thetas).astype(np.float32)).to(device)
    return rc, rs


def alloc_graph(cfg: N.EnvCfg, B: int, device) -> Dict[str, torch.Tensor]:
    Nn, E = cfg.num_nodes, cfg.num_edges
    return dict(
        nodes=torch.empty(B, Nn, cfg.node_dim, device=device),
        edges=torch.empty(B, E, 4, device=device),
        states=torch.empty(B, Nn, cfg.state_dim, device=device),
        receivers=torch.empty(B, E, dtype=torch.int32, device=device),
        senders=torch.empty(B, E, dtype=torch.int32, device=device),
        node_type=torch.empty(B, Nn, dtype=torch.int32, device=device),
        n_node=torch.empty(B, dtype=torch.int32, device=device),
        n_edge=torch.empty(B, dtype=torch.int32, device=device),
    )


def _graph_out(cfg: N.EnvCfg, g: Dict[str, torch.Tensor], B: int) -> N.GraphOut:
    Nn, E = cfg.num_nodes, cfg.num_edges
    N.expect_shape(g["nodes"], (B, Nn, cfg.node_dim), "graph.nodes")
    N.expect_shape(g["edges"], (B, E, 4), "graph.edges")
    N.expect_shape(g["states"], (B, Nn, cfg.state_dim), "graph.states")
    N.expect_shape(g["receivers"], (B, E), "graph.receivers")
    N.expect_shape(g["senders"], (B, E), "graph.senders")
    N.expect_shape(g["node_type"], (B, Nn), "graph.node_type")
    N.expect_shape(g["n_node"], (B,), "graph.n_node")
    N.expect_shape(g["n_edge"], (B,), "graph.n_edge")
    go = N.GraphOut()
    go.nodes = N.ptr(g["nodes"], name="graph.nodes")
    go.edges = N.ptr(g["edges"], name="graph.edges")
    go.states = N.ptr(g["states"], name="graph.states")
    go.receivers = N.ptr(g["receivers"], torch.int32, "graph.receivers")
    go.senders = N.ptr(g["senders"], torch.int32, "graph.senders")
    go.node_type = N.ptr(g["node_type"], torch.int32, "graph.node_type")
    go.n_node = N.ptr(g["n_node"], torch.int32, "graph.n_node")
    go.n_edge = N.ptr(g["n_edge"], torch.int32, "graph.n_edge")
    return go


def _check_state(cfg: N.EnvCfg, agent, goal, obst, hits, B):
    n, sd = cfg.n_agents, cfg.state_dim
    N.expect_shape(agent, (B, n, sd), "agent")
    N.expect_shape(goal, (B, cfg.n_goals, sd), "goal")
    if cfg.n_obs > 0:
        if obst is None:
            raise ValueError("obst is required when n_obs > 0")
        N.expect_shape(obst, (B, cfg.n_obs, cfg.obst_stride), "obst")
    if hits is not None:
        N.expect_shape(hits, (B, n, cfg.top_k, 2), "hits")


def env_step(cfg: N.EnvCfg, agent, action, goal, obst, hits, ray_cos, ray_sin,
             next_agent, next_hits, reward, cost, graph: Optional[Dict[str, torch.Tensor]] = None):
    """dgppo_env_step.  action=None -> sense-only (graph of the given state)."""
    B = agent.shape[0]
    _check_state(cfg, agent, goal, obst, hits, B)
    n = cfg.n_agents
    if action is not None:
        N.expect_shape(action, (B, n, 2), "action")
        N.expect_shape(reward, (B,), "reward")
        N.expect_shape(cost, (B, n, cfg.n_cost), "cost")
    if next_agent is not None:
        N.expect_shape(next_agent, (B, n, cfg.state_dim), "next_agent")
    if next_hits is not None:
        N.expect_shape(next_hits, (B, n, cfg.top_k, 2), "next_hits")
    if cfg.is_lidar and cfg.n_obs > 0:
        N.expect_shape(ray_cos, (cfg.n_rays,), "ray_cos")
        N.expect_shape(ray_sin, (cfg.n_rays,), "ray_sin")
    go = _graph_out(cfg, graph, B) if graph is not None else None
    rc = N.lib().dgppo_env_step(
        C.byref(cfg), N.ptr(agent, name="agent"), N.ptr(action, name="action"), N.ptr(goal, name="goal"),
        N.ptr(obst, name="obst"), N.ptr(hits, name="hits"), N.ptr(ray_cos, name="ray_cos"), N.ptr(ray_sin, name="ray_sin"),
        N.ptr(next_agent, name="next_agent"), N.ptr(next_hits, name="next_hits"), N.ptr(reward, name="reward"),
        N.ptr(cost, name="cost"), C.byref(go) if go is not None else None, C.c_int32(B), N.stream_ptr())
    N.check(rc, "dgppo_env_step")


def graph_materialize(cfg: N.EnvCfg, agent, goal, obst, hits, graph: Dict[str, torch.Tensor]):
    B = agent.shape[0]
    _check_state(cfg, agent, goal, obst, hits, B)
    go = _graph_out(cfg, graph, B)
    rc = N.lib().dgppo_graph_materialize(
        C.byref(cfg), N.ptr(agent, name="agent"), N.ptr(goal, name="goal"), N.ptr(obst, name="obst"),
        N.ptr(hits, name="hits"), C.byref(go), C.c_int32(B), N.stream_ptr())
    N.check(rc, "dgppo_graph_materialize")


def env_reset(cfg: N.EnvCfg, seeds: torch.Tensor, agent, goal, obst, n_failed: torch.Tensor = None):
    """n_failed: optional int32 device counter (caller-zeroed), += 1 per env whose bounded rejection loops ran out — read it at
    the next host sync and do not use the batch when it is non-zero (dgppo_env_reset_checked)."""
    B = seeds.shape[0]
    n, sd = cfg.n_agents, cfg.state_dim
    N.expect_shape(agent, (B, n, sd), "agent")
    N.expect_shape(goal, (B, cfg.n_goals, sd), "goal")
    if cfg.n_obs > 0:
        N.expect_shape(obst, (B, cfg.n_obs, cfg.obst_stride), "obst")
    if n_failed is not None:
        N.expect_shape(n_failed, (1,), "n_failed")
    rc = N.lib().dgppo_env_reset_checked(C.byref(cfg), N.ptr(seeds, torch.int64, "seeds"), N.ptr(agent, name="agent"),
                                         N.ptr(goal, name="goal"), N.ptr(obst, name="obst"),
                                         N.ptr(n_failed, torch.int32, "n_failed"), C.c_int32(B), N.stream_ptr())
    N.check(rc, "dgppo_env_reset")


def randn_rows(seed: int, out: torch.Tensor, global_row_len: int, col_offset: int):
    """out [rows, row_len] = the column window [col_offset, col_offset + row_len) of randn(seed) viewed as rows of
    global_row_len (data-parallel rollouts: a rank's share of the global noise)."""
    rows = int(out.shape[0])
    row_len = out.numel() // max(rows, 1)
    rc = N.lib().dgppo_randn_rows(C.c_uint64(seed & 0xFFFFFFFFFFFFFFFF), N.ptr(out, name="out"), C.c_int64(rows),
                                  C.c_int64(row_len), C.c_int64(global_row_len), C.c_int64(col_offset), N.stream_ptr())
    N.check(rc, "dgppo_randn_rows")


def randn(seed: int, offset: int, out: torch.Tensor):
    rc = N.lib().dgppo_randn(C.c_uint64(seed & 0xFFFFFFFFFFFFFFFF), C.c_uint64(offset), N.ptr(out, name="out"),
                             C.c_int64(out.numel()), N.stream_ptr())
    N.check(rc, "dgppo_randn")
