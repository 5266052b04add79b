"""dgppo/trainer/utils.py counterparts that are host logic: connectivity probe and rollout wrappers."""
from __future__ import annotations


def is_connected() -> bool:
    """The reference dials 8.8.8.8:53 (dgppo/trainer/utils.py:133-149); this build never touches the network
     and always reports offline."""
    return False


def rollout(env, algo, keys):
    """stochastic rollouts of `algo` on `env` for the given seeds (dgppo/trainer/utils.py:22-57), batched."""
    return algo.collect(algo.params, keys)


def test_rollout(env, algo, keys):
    """deterministic rollouts (dgppo/trainer/utils.py:60-86), batched."""
    return algo.collect_deterministic(keys, env=env)
