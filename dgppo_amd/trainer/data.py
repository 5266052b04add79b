"""Rollout record with the reference's field names (dgppo/trainer/data.py:8-32).  `graph` / `next_graph` are lazy views:
the engine stores compact records and materialises GraphsTuples with a HIP kernel only when asked."""
from __future__ import annotations

from typing import Any, NamedTuple, Optional


class Rollout(NamedTuple):
    graph: Any
    actions: Any
    rnn_states: Any
    rewards: Any
    costs: Any
    dones: Any
    log_pis: Optional[Any]
    next_graph: Any

    @property
    def length(self) -> int:
        return self.rewards.shape[0]

    @property
    def time_horizon(self) -> int:
        return self.rewards.shape[1]

    @property
    def num_agents(self) -> int:
        return self.costs.shape[2]

    @property
    def n_data(self) -> int:
        return self.length * self.time_horizon
