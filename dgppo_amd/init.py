"""Host-side parameter initialisation (construction time only, not on the hot path): the distributions the reference
uses — flax orthogonal(scale) Dense kernels, zero biases, LayerNorm (1, 0), GRUCell lecun-normal input kernels and
orthogonal recurrent kernels (dgppo/nn/utils.py:20-27).  JAX's PRNG stream cannot be reproduced, so the
values differ from a JAX run with the same seed; the distributions do not."""
from __future__ import annotations

import math

import numpy as np


def orthogonal(rng: np.random.Generator, n_in: int, n_out: int, scale: float = 1.0) -> np.ndarray:
    a = rng.standard_normal((max(n_in, n_out), min(n_in, n_out)))
    q, r = np.linalg.qr(a)
    q = q * np.sign(np.diagonal(r))
    if n_in < n_out:
        q = q.T
    return (scale * q).astype(np.float32)


def lecun_normal(rng: np.random.Generator, n_in: int, n_out: int) -> np.ndarray:
    std = math.sqrt(1.0 / n_in) / 0.87962566103423978
    x = rng.standard_normal((n_in, n_out))
    bad = np.abs(x) > 2.0
    while bad.any():                      # truncated normal on [-2, 2]
        x[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(x) > 2.0
    return (x * std).astype(np.float32)


def _dense(rng, n_in, n_out, bias=True, scale=1.0):
    p = {"kernel": orthogonal(rng, n_in, n_out, scale)}
    if bias:
        p["bias"] = np.zeros(n_out, np.float32)
    return p


def _gnn(rng, node_dim, n_layers, msg_dim=32, out_dim=64, n_heads=3):
    p, f = {}, node_dim
    for i in range(n_layers):
        d = out_dim if i == n_layers - 1 else msg_dim
        hd = d * n_heads
        p[f"GraphTransformer_{i}"] = {"Dense_0": _dense(rng, f, hd), "Dense_1": _dense(rng, f, hd), "Dense_2": _dense(rng, f, hd),
                                      "Dense_3": _dense(rng, 4, hd, bias=False), "Dense_4": _dense(rng, f, d)}
        f = d
    return p


def _mlp(rng, f_in: int = 64):
    p = {}
    for i in range(2):
        p[f"Dense_{i}"] = _dense(rng, f_in if i == 0 else 64, 64)
        p[f"LayerNorm_{i}"] = {"scale": np.ones(64, np.float32), "bias": np.zeros(64, np.float32)}
    return p


def _gru(rng):
    z = lambda: np.zeros(64, np.float32)
    return {"ir": {"kernel": lecun_normal(rng, 64, 64), "bias": z()}, "iz": {"kernel": lecun_normal(rng, 64, 64), "bias": z()},
            "in": {"kernel": lecun_normal(rng, 64, 64), "bias": z()}, "hr": {"kernel": orthogonal(rng, 64, 64)},
            "hz": {"kernel": orthogonal(rng, 64, 64)}, "hn": {"kernel": orthogonal(rng, 64, 64), "bias": z()}}


def _lstm(rng):
    """flax nn.LSTMCell(64): input Denses without bias (lecun-normal), hidden Denses with bias (orthogonal)"""
    p = {}
    for g in "ifgo":
        p["i" + g] = {"kernel": lecun_normal(rng, 64, 64)}
        p["h" + g] = {"kernel": orthogonal(rng, 64, 64), "bias": np.zeros(64, np.float32)}
    return p


def _rnn(rng, rnn_layers: int, lstm: bool = False) -> dict:
    """RNN_0 of dgppo/nn/rnn.py:14-30: layer l's cell is auto-named GRUCell_{2l+1} / LSTMCell_{3l+2} (one / two
    isinstance probes instantiate the class before the instance in use)"""
    if lstm:
        return {f"LSTMCell_{3 * l + 2}": _lstm(rng) for l in range(rnn_layers)}
    return {f"GRUCell_{2 * l + 1}": _gru(rng) for l in range(rnn_layers)}


def init_policy(seed: int, node_dim: int, action_dim: int, gnn_layers: int, rnn_layers: int = 1, lstm: bool = False) -> dict:
    """rnn_layers = 0: --no-rnn (no RNN_0 entry)"""
    rng = np.random.default_rng([seed, 1])
    base = {"GraphTransformerGNN_0": _gnn(rng, node_dim, gnn_layers), "PolicyGNNHead": _mlp(rng)}
    if rnn_layers > 0:
        base["RNN_0"] = _rnn(rng, rnn_layers, lstm)
    return {"params": {
        "PolicyNet_0": base,
        "ScaleHid": _dense(rng, 64, 64, scale=0.01),
        "OutputDenseMean": _dense(rng, 64, action_dim),
        "OutputDenseStdTrans": _dense(rng, 64, action_dim)}}


def init_value(seed: int, node_dim: int, n_out: int, gnn_layers: int, stream: int, global_info: bool = False,
               rnn_layers: int = 1, lstm: bool = False) -> dict:
    """global_info: DecRStateFn(use_global_info=True) — the head's first Dense takes [x_i | mean_j x_j] (value.py:66-68)"""
    rng = np.random.default_rng([seed, stream])
    return {"params": {"GraphTransformerGNN_0": _gnn(rng, node_dim, gnn_layers), "ValueGNNHead": _mlp(rng, 128 if global_info else 64),
                       **({"RNN_0": _rnn(rng, rnn_layers, lstm)} if rnn_layers > 0 else {}), "Dense_0": _dense(rng, 64, n_out)}}
