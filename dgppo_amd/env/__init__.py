"""dgppo.env equivalent (dgppo/env/__init__.py:9-53): make_env + the environment families on the hot path."""
from typing import Optional

from .base import MultiAgentEnv
from .envs import LidarSpread, LidarTarget, LidarBicycleTarget, MPESpread, MPETarget, LidarEnvState, MPEEnvState, Rectangle

ENV = {
    "MPETarget": MPETarget, "MPESpread": MPESpread, "LidarSpread": LidarSpread, "LidarTarget": LidarTarget,
    "LidarBicycleTarget": LidarBicycleTarget,
}
# registered in the reference but outside the hot-path scope of this build
_NOT_BUILT = ("MPELine", "MPEFormation", "MPECorridor", "MPEConnectSpread", "LidarLine", "VMASReverseTransport", "VMASWheel")

DEFAULT_MAX_STEP = 128


def make_env(env_id: str, num_agents: int, max_step: int = None, full_observation: bool = False,
             num_obs: Optional[int] = None, n_rays: Optional[int] = None) -> MultiAgentEnv:
    if env_id in _NOT_BUILT:
        raise NotImplementedError(f"{env_id} is registered by the reference but not part of the MI355X hot-path build "
                                  f"(available: {sorted(ENV)})")
    assert env_id in ENV.keys(), f"Environment {env_id} not implemented."
    params = dict(ENV[env_id].PARAMS)      # a copy: the reference mutates the shared class dict
    max_step = DEFAULT_MAX_STEP if max_step is None else max_step
    if num_obs is not None:
        params["n_obs"] = num_obs
    if n_rays is not None and "n_rays" in params:
        params["n_rays"] = n_rays
    if full_observation:
        params["comm_radius"] = params["default_area_size"] * 10
    return ENV[env_id](num_agents=num_agents, area_size=None, max_step=max_step, dt=0.03, params=params)
