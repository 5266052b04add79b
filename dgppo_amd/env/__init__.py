"""Environment registry and `make_env` with the reference's signature (dgppo/env/__init__.py:29-53)."""
from typing import Optional

from . import envs as _envs
from .base import MultiAgentEnv
from .envs import LidarEnvState, MPEEnvState, Rectangle          # re-exported: users import them from dgppo.env

DEFAULT_MAX_STEP = 128
_BUILT = ("MPETarget", "MPESpread", "MPELine", "MPEFormation", "MPECorridor", "MPEConnectSpread", "LidarSpread",
          "LidarTarget", "LidarLine", "LidarBicycleTarget")
# registered by the reference, outside the scope of this build
_REFERENCE_ONLY = ("VMASReverseTransport", "VMASWheel")
ENV = {name: getattr(_envs, name) for name in _BUILT}
globals().update(ENV)                                              # `from dgppo.env import LidarSpread` keeps working


def make_env(env_id: str, num_agents: int, max_step: int = None, full_observation: bool = False,
             num_obs: Optional[int] = None, n_rays: Optional[int] = None) -> MultiAgentEnv:
    if env_id in _REFERENCE_ONLY:
        raise NotImplementedError(f"{env_id} is registered by the reference but not part of the MI355X hot-path build "
                                  f"(available: {sorted(ENV)})")
    if env_id not in ENV:
        raise AssertionError(f"Environment {env_id} not implemented.")
    cls = ENV[env_id]
    # a private copy of the class-level PARAMS: the reference writes the overrides into the shared dict, so its train and
    # test envs alias each other
    overrides = {"n_obs": num_obs, "n_rays": n_rays if "n_rays" in cls.PARAMS else None,
                 "comm_radius": cls.PARAMS["default_area_size"] * 10 if full_observation else None}
    params = {**cls.PARAMS, **{k: v for k, v in overrides.items() if v is not None}}
    return cls(num_agents=num_agents, area_size=None, dt=0.03, params=params,
               max_step=DEFAULT_MAX_STEP if max_step is None else max_step)
