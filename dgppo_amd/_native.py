"""ctypes binding of libdgppo_hip.so (C ABI: include/dgppo_hip.h).

The product path has NO CPU fallback: if the HIP library is missing this module raises at import
of the first symbol, and every wrapper refuses non-CUDA tensors.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# DGPPO_HIP_LIB: developer override (the stamps build of tools/stamps_wave.py); the default is the in-tree library
LIB_PATH = os.environ.get("DGPPO_HIP_LIB") or os.path.join(_HERE, "csrc", "libdgppo_hip.so")

ABI_VERSION = 3
OPT_STATE_FLOATS = 8 + 2 * 256     # DGPPO_OPT_STATE_FLOATS (include/dgppo_hip.h)

ENV_KINDS = {"LidarSpread": 0, "LidarTarget": 1, "LidarBicycleTarget": 2, "MPESpread": 3, "MPETarget": 4,
             "LidarLine": 5, "MPELine": 6, "MPEFormation": 7, "MPECorridor": 8, "MPEConnectSpread": 9}
GOALS_NODES, GOALS_LINE, GOALS_LINE_INTERIOR, GOALS_CIRCLE = range(4)
RECT_STRIDE = 16


class EnvCfg(C.Structure):
    """struct dgppo_env_cfg"""
    _fields_ = [
        ("kind", C.c_int32), ("n_agents", C.c_int32), ("n_goals", C.c_int32), ("n_obs", C.c_int32),
        ("n_rays", C.c_int32), ("top_k", C.c_int32), ("state_dim", C.c_int32), ("node_dim", C.c_int32),
        ("area_size", C.c_float), ("dt", C.c_float), ("car_radius", C.c_float), ("comm_radius", C.c_float),
        ("obs_radius", C.c_float), ("dist2goal", C.c_float), ("two_car_radius", C.c_float),
        ("lidar_mask_radius", C.c_float), ("eye_offset", C.c_float), ("car_plus_obs", C.c_float),
        ("vel_limit", C.c_float), ("reset_min_dist", C.c_float),
        ("reward_goals", C.c_int32), ("n_cost", C.c_int32), ("obs_mask_radius", C.c_float), ("y_limit", C.c_float),
        ("connect_radius", C.c_float), ("reset_side_y", C.c_float), ("goal_shift_y", C.c_float), ("line_min_dist", C.c_float),
    ]

    # ---- derived sizes (mirror csrc/common.h) ----
    @property
    def is_lidar(self):
        return self.kind <= 2 or self.kind == 5

    @property
    def is_spread(self):
        return self.kind not in (1, 2, 4)

    @property
    def obs_nodes(self):
        if self.is_lidar:
            return self.n_agents * self.top_k if self.n_obs > 0 else 0
        return self.n_obs

    @property
    def obs_slots(self):
        if self.is_lidar:
            return self.top_k if self.n_obs > 0 else 0
        return self.n_obs

    @property
    def goal_slots(self):
        return self.n_goals if self.is_spread else 1

    @property
    def num_nodes(self):
        return self.n_agents + self.n_goals + self.obs_nodes + 1

    @property
    def num_edges(self):
        return self.n_agents * (self.n_agents + self.goal_slots + self.obs_slots)

    @property
    def fan_in(self):
        return self.n_agents + self.goal_slots + self.obs_slots

    @property
    def obst_stride(self):
        return RECT_STRIDE if self.is_lidar else self.state_dim


def make_env_cfg(kind: int, n_agents: int, n_obs: int, n_rays: int = 32, top_k: int = 8, area_size: float = None,
                 dt: float = 0.03, car_radius: float = 0.05, comm_radius: float = 0.5, obs_radius: float = None,
                 dist2goal: float = 0.01, connect_radius: float = 0.45, corridor_width: float = 0.2) -> EnvCfg:
    """Thresholds are formed in Python doubles then rounded to fp32, exactly as the reference's weakly-typed
    Python floats are (e.g. `comm_radius - 1e-1`, dgppo/env/lidar_env/lidar_spread.py:88).  area_size / obs_radius default
    to the PARAMS of the kind (1.5 / 0.05; MPECorridor and MPEConnectSpread: area 1.0, obs_radius derived / 0.25)."""
    is_lidar = kind <= 2 or kind == 5
    corridor, connect = kind == 8, kind == 9
    if area_size is None:
        area_size = 1.0 if (corridor or connect) else 1.5
    if corridor:
        n_obs = 2                                                          # mpe_corridor.py:35-37
        obs_radius = (area_size - corridor_width) / 4                      # :39
    elif connect:
        n_obs = 1                                                          # mpe_connect_spread.py:38-40
        obs_radius = 0.25 if obs_radius is None else obs_radius
    elif obs_radius is None:
        obs_radius = 0.05
    sd = 5 if kind == 2 else 4
    c = EnvCfg()
    c.kind, c.n_agents, c.n_obs = kind, n_agents, n_obs
    c.n_goals = 2 if kind in (5, 6) else (1 if kind == 7 else n_agents)
    c.n_rays, c.top_k = (n_rays, top_k) if is_lidar else (0, 0)
    c.state_dim, c.node_dim = sd, sd + 3
    c.area_size, c.dt, c.car_radius, c.comm_radius = area_size, dt, car_radius, comm_radius
    c.obs_radius, c.dist2goal = obs_radius, dist2goal
    c.two_car_radius = car_radius * 2
    c.lidar_mask_radius = comm_radius - 1e-1
    c.eye_offset = comm_radius + 1
    c.car_plus_obs = car_radius + obs_radius
    c.vel_limit = 0.5 if is_lidar else 1.0
    c.reset_min_dist = (2.2 * car_radius) if (is_lidar and kind != 5) else ((2.3 if connect else 2) * car_radius)
    # ---- task variants ----
    c.reward_goals = {5: GOALS_LINE, 6: GOALS_LINE if n_agents > 3 else GOALS_LINE_INTERIOR, 7: GOALS_CIRCLE}.get(kind, GOALS_NODES)
    c.n_cost = 3 if connect else 2
    c.obs_mask_radius = comm_radius * 100 if (corridor or connect) else comm_radius
    c.y_limit = area_size * 2 if (corridor or connect) else area_size
    c.connect_radius = connect_radius
    c.reset_side_y = ((area_size - obs_radius * 2) / 2 - 1.5 * car_radius) if (corridor or connect) else area_size
    c.goal_shift_y = (area_size - (area_size - obs_radius * 2) / 2 + 1.5 * car_radius) if (corridor or connect) else 0.0
    if kind in (5, 6):
        c.line_min_dist = n_agents * 5 * car_radius if (kind == 6 and n_agents <= 3) else (n_agents - 2) * 6 * car_radius
    else:
        c.line_min_dist = 0.0
    return c


class GraphOut(C.Structure):
    """struct dgppo_graph_out"""
    _fields_ = [(k, C.c_void_p) for k in
                ("nodes", "edges", "states", "receivers", "senders", "node_type", "n_node", "n_edge")]


_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    """Load the HIP library or fail loudly (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"dgppo_amd: HIP library not built: {LIB_PATH} is missing. "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` (or `make -C dgppo_amd/csrc`). "
            "There is no CPU fallback.")
    l = C.CDLL(LIB_PATH)
    l.dgppo_abi_version.restype = C.c_int32
    l.dgppo_last_error.restype = C.c_char_p
    v = l.dgppo_abi_version()
    if v != ABI_VERSION:
        raise RuntimeError(f"dgppo_amd: ABI mismatch: library {v}, python {ABI_VERSION}")
    _lib = l
    return l


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().dgppo_last_error().decode("utf-8", "replace")
        if rc > 0:
            raise RuntimeError(f"{what}: hipError {rc}: {msg}")
        raise ValueError(f"{what}: {msg}")


def ptr(t: Optional[torch.Tensor], dtype=torch.float32, name: str = "tensor") -> C.c_void_p:
    """device pointer of a contiguous CUDA tensor (None -> NULL)."""
    if t is None:
        return C.c_void_p(0)
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU (the HIP path has no CPU fallback)")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    return C.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def stream_ptr() -> C.c_void_p:
    """HIP stream the kernels are enqueued on = torch's current stream.  Called once per launch, so it goes through the two
    C entry points directly; `torch.cuda.current_stream().cuda_stream` costs ~15 us of Python per call (measured)."""
    if _raw_stream is not None and _raw_device is not None:
        return C.c_void_p(_raw_stream(_raw_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def expect_shape(t: torch.Tensor, shape, name: str):
    if tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
