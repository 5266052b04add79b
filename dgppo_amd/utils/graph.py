"""GraphsTuple container with the reference's field order (dgppo/utils/graph.py:47-189).  Arrays are torch tensors
(GPU) produced by the HIP kernels; a leading batch axis is allowed (the reference vmaps single graphs instead)."""
from __future__ import annotations

from typing import Any, NamedTuple, Optional


class GraphsTuple(NamedTuple):
    n_node: Any
    n_edge: Any
    nodes: Any
    edges: Any
    states: Any
    receivers: Any
    senders: Any
    node_type: Any
    env_states: Any
    connectivity: Any = None

    @property
    def is_single(self) -> bool:
        return self.n_node.ndim == 0

    @property
    def batch_shape(self):
        return tuple(self.n_node.shape)

    def type_nodes(self, type_idx: int, n_type: int):
        """rows of one node type (graph.py:115-127: with the static node order this is a slice)."""
        return _type_rows(self.nodes, self.node_type, type_idx, n_type)

    def type_states(self, type_idx: int, n_type: int):
        return _type_rows(self.states, self.node_type, type_idx, n_type)

    def without_edge(self):
        return self._replace(edges=None)


def _type_rows(x, node_type, type_idx, n_type):
    nt = node_type if node_type.ndim == 1 else node_type.reshape(-1, node_type.shape[-1])[0]
    idx = (nt == type_idx).nonzero().flatten()
    assert idx.numel() == n_type, f"expected {n_type} nodes of type {type_idx}, found {idx.numel()}"
    start = int(idx[0]) if n_type > 0 else 0
    return x[..., start:start + n_type, :]
