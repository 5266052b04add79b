"""What every algorithm of this build exposes — the reference's `Algorithm` interface (dgppo/algo/base.py:8-99): four
read-only dimensions, `config` / `params`, and act / step / collect / update / save / load."""
import abc


def _readonly(name: str):
    return property(lambda self: getattr(self, "_" + name), doc=f"{name} given at construction")


class Algorithm(abc.ABC):
    node_dim, edge_dim, action_dim, n_agents = (_readonly(k) for k in ("node_dim", "edge_dim", "action_dim", "n_agents"))

    def __init__(self, env, node_dim: int, edge_dim: int, action_dim: int, n_agents: int):
        self._env = env
        self._node_dim, self._edge_dim, self._action_dim, self._n_agents = node_dim, edge_dim, action_dim, n_agents
        self.init_rnn_state = None   # set by the subclasses: zeros of shape (rnn_layers, n_agents, 1, 64)

    # -- to be provided by an algorithm ----------------------------------------------------------------------------
    config = abc.abstractproperty(doc="hyper-parameters written to config.yaml")
    params = abc.abstractproperty(doc="flax-named parameter trees")

    @abc.abstractmethod
    def act(self, graph, rnn_state, params=None):
        """deterministic action of one graph -> (action, new_rnn_state)"""

    @abc.abstractmethod
    def step(self, graph, rnn_state, key, params=None):
        """sampled action of one graph -> (action, log_pi, new_rnn_state)"""

    @abc.abstractmethod
    def collect(self, params, key):
        """one training rollout per key -> Rollout"""

    @abc.abstractmethod
    def update(self, rollout, step: int) -> dict:
        """one training iteration on a collected rollout -> logged scalars"""

    @abc.abstractmethod
    def save(self, save_dir: str, step: int):
        """write {save_dir}/{step}/*.pkl"""

    @abc.abstractmethod
    def load(self, load_dir: str, step: int):
        """read {load_dir}/{step}/*.pkl"""
