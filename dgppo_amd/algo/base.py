"""Algorithm interface of the reference (dgppo/algo/base.py:8-99)."""
from abc import ABC, abstractmethod


class Algorithm(ABC):
    def __init__(self, env, node_dim: int, edge_dim: int, action_dim: int, n_agents: int):
        self._env = env
        self._node_dim = node_dim
        self._edge_dim = edge_dim
        self._action_dim = action_dim
        self._n_agents = n_agents
        self.init_rnn_state = None

    @property
    def node_dim(self) -> int:
        return self._node_dim

    @property
    def edge_dim(self) -> int:
        return self._edge_dim

    @property
    def action_dim(self) -> int:
        return self._action_dim

    @property
    def n_agents(self) -> int:
        return self._n_agents

    @property
    @abstractmethod
    def config(self) -> dict: ...

    @property
    @abstractmethod
    def params(self): ...

    @abstractmethod
    def act(self, graph, rnn_state, params=None): ...

    @abstractmethod
    def step(self, graph, rnn_state, key, params=None): ...

    @abstractmethod
    def collect(self, params, key): ...

    @abstractmethod
    def update(self, rollout, step: int) -> dict: ...

    @abstractmethod
    def save(self, save_dir: str, step: int): ...

    @abstractmethod
    def load(self, load_dir: str, step: int): ...
