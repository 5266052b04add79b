"""dgppo/trainer/utils.py counterparts that are host logic: the connectivity probe and the single-environment rollout
functions with the reference's signatures.

`rollout(env, actor, init_rnn_state, key)` / `test_rollout(env, actor, init_rnn_state, key, stochastic=False)` drive ONE
environment through `env.reset` / `env.step` and an arbitrary `actor` callable, exactly like the reference's `lax.scan`
bodies (dgppo/trainer/utils.py:22-86): they are the API-level path (each step goes through the materialised GraphsTuple).
Training does not use them — `algo.collect` runs all environments at once on the compact record (`rollout_batch` /
`test_rollout_batch` below) — but anything written against the reference's helpers keeps working.

Keys are integer seeds (JAX threefry streams cannot be reproduced); `_split` derives child seeds the way
`jax.random.split` derives child keys: deterministically and without overlap."""
from __future__ import annotations

from typing import Callable

import numpy as np
import torch

from ..utils.graph import GraphsTuple
from .data import Rollout


def is_connected() -> bool:
    """The reference dials 8.8.8.8:53 (dgppo/trainer/utils.py:133-149); this build never touches the network
     and always reports offline."""
    return False


def _seed_of(key) -> int:
    if torch.is_tensor(key):
        key = key.detach().cpu().numpy()
    return int(np.asarray(key).reshape(-1)[-1]) & 0x7FFFFFFFFFFFFFFF


def _split(key, num: int):
    """`jax.random.split(key, num)` for integer seeds: `num` child seeds from a SeedSequence keyed by the parent."""
    ss = np.random.SeedSequence(_seed_of(key))
    return [int(s.generate_state(1, dtype=np.uint64)[0] >> np.uint64(1)) for s in ss.spawn(num)]


def _stack_tree(items):
    first = items[0]
    if first is None:
        return None
    if torch.is_tensor(first):
        return torch.stack(items, 0)
    if isinstance(first, tuple) and hasattr(first, "_fields"):
        return type(first)(*[_stack_tree([it[i] for it in items]) for i in range(len(first))])
    return first


def _stack_graphs(graphs) -> GraphsTuple:
    """[T] single graphs -> one GraphsTuple with a leading time axis (what `lax.scan` stacking produces)."""
    return GraphsTuple(*[_stack_tree([getattr(g, f) for g in graphs]) for f in GraphsTuple._fields])


def rollout(env, actor: Callable, init_rnn_state, key) -> Rollout:
    """dgppo/trainer/utils.py:22-57.  actor: (graph, rnn_state, key) -> (action, log_pi, new_rnn_state).
    The stored `rnn_states[t]` is the carry BEFORE step t (utils.py:46-51)."""
    key_x0, _key_z0, key = _split(key, 3)
    graph = env.reset(key_x0)
    rnn_state = init_rnn_state
    T = env.max_episode_steps
    keys = _split(key, T)
    graphs, actions, rnn_states, rewards, costs, dones, log_pis, next_graphs = [], [], [], [], [], [], [], []
    for t in range(T):
        action, log_pi, new_rnn_state = actor(graph, rnn_state, keys[t])
        next_graph, reward, cost, done, _info = env.step(graph, action)
        graphs.append(graph); actions.append(action); rnn_states.append(rnn_state); rewards.append(reward)
        costs.append(cost); dones.append(done); log_pis.append(log_pi); next_graphs.append(next_graph)
        graph, rnn_state = next_graph, new_rnn_state
    st = lambda xs: torch.stack([torch.as_tensor(x) for x in xs], 0)
    return Rollout(_stack_graphs(graphs), st(actions), st(rnn_states), st(rewards), st(costs), st(dones), st(log_pis),
                   _stack_graphs(next_graphs))


def test_rollout(env, actor: Callable, init_rnn_state, key, stochastic: bool = False) -> Rollout:
    """dgppo/trainer/utils.py:60-86.  actor: (graph, rnn_state[, key]) -> (action, new_rnn_state).
    The stored `rnn_states[t]` is the carry AFTER the actor call of step t (utils.py:71-77);
    `log_pis` is None."""
    key_x0, key = _split(key, 2)
    graph = env.reset(key_x0)
    rnn_state = init_rnn_state
    T = env.max_episode_steps
    keys = _split(key, T)
    graphs, actions, rnn_states, rewards, costs, dones, next_graphs = [], [], [], [], [], [], []
    for t in range(T):
        if not stochastic:
            action, rnn_state = actor(graph, rnn_state)
        else:
            action, rnn_state = actor(graph, rnn_state, keys[t])
        next_graph, reward, cost, done, _info = env.step(graph, action)
        graphs.append(graph); actions.append(action); rnn_states.append(rnn_state); rewards.append(reward)
        costs.append(cost); dones.append(done); next_graphs.append(next_graph)
        graph = next_graph
    st = lambda xs: torch.stack([torch.as_tensor(x) for x in xs], 0)
    return Rollout(_stack_graphs(graphs), st(actions), st(rnn_states), st(rewards), st(costs), st(dones), None,
                   _stack_graphs(next_graphs))


test_rollout.__test__ = False   # not a pytest test despite the reference's name


def rollout_batch(env, algo, keys) -> Rollout:
    """all environments at once on the compact record: what `jax.vmap(rollout)` is in the reference (informarl.py:177-186)"""
    return algo.collect(algo.params, keys)


def test_rollout_batch(env, algo, keys) -> Rollout:
    """batched deterministic rollouts (trainer.py:98-100 vmaps test_rollout over the fixed test keys)"""
    return algo.collect_deterministic(keys, env=env)


test_rollout_batch.__test__ = False
