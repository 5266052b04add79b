"""The rollout record handed between `algo.collect`, `algo.update` and the trainer.  Field names and order are API: they
are the reference's (dgppo/trainer/data.py:8-16).  `graph` / `next_graph` are lazy views here — the engine stores compact
records and materialises GraphsTuples with a HIP kernel only when someone reads them."""
from __future__ import annotations

import collections

_FIELDS = ("graph", "actions", "rnn_states", "rewards", "costs", "dones", "log_pis", "next_graph")


class Rollout(collections.namedtuple("Rollout", _FIELDS)):
    """[B, T, ...] arrays; `log_pis` may be None (deterministic rollouts)."""
    __slots__ = ()

    def _dim(self, field: str, axis: int) -> int:
        return int(getattr(self, field).shape[axis])

    # the four size helpers of the reference (data.py:18-32)
    length = property(lambda self: self._dim("rewards", 0), doc="number of environments B")
    time_horizon = property(lambda self: self._dim("rewards", 1), doc="steps per environment T")
    num_agents = property(lambda self: self._dim("costs", 2), doc="agents per environment")
    n_data = property(lambda self: self._dim("rewards", 0) * self._dim("rewards", 1), doc="B * T")
