"""
ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product path (dgppo_amd/).

Torch-CPU fp32 restatement of the reference's networks in the reference's own (per-edge, gather / segment-softmax /
segment-sum) form, with autograd available for gradient checks.  Citations are file:line under /root/reference.

PARITY UNPINNED: flax / jraph / tensorflow-probability / optax are absent from the build container, so the
library semantics restated here ([upstream] in SURVEY App. A.5-A.7, A.11) are pinned only by the analytic
known-answer tests in tests/test_oracle_nn.py, not by outputs of the reference.

Parameter trees use the flax auto-names of SURVEY A.9 (Dense kernels are [in, out]).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
LOG_2PI = math.log(2.0 * math.pi)


# ----------------------------------------------------------------------------------------------------------------------
# initialisers (dgppo/nn/utils.py:20-27; flax defaults for GRUCell)
# ----------------------------------------------------------------------------------------------------------------------
def orthogonal(gen: torch.Generator, n_in: int, n_out: int, scale: float = 1.0) -> Tensor:
    """flax.linen.initializers.orthogonal(scale) for a 2-D [in, out] kernel: QR of a normal matrix, sign-fixed."""
    rows, cols = (n_in, n_out)
    a = torch.randn(max(rows, cols), min(rows, cols), generator=gen, dtype=torch.float64)
    q, r = torch.linalg.qr(a)
    q = q * torch.sign(torch.diagonal(r))
    if rows < cols:
        q = q.T
    return (scale * q).to(torch.float32).contiguous()


def lecun_normal(gen: torch.Generator, n_in: int, n_out: int) -> Tensor:
    std = math.sqrt(1.0 / n_in) / 0.87962566103423978  # truncated-normal correction used by jax variance_scaling
    x = torch.empty(n_in, n_out, dtype=torch.float64)
    torch.nn.init.trunc_normal_(x, mean=0.0, std=1.0, a=-2.0, b=2.0, generator=gen)
    return (x * std).to(torch.float32)


def _dense(gen, n_in, n_out, bias=True, scale=1.0):
    p = {"kernel": orthogonal(gen, n_in, n_out, scale)}
    if bias:
        p["bias"] = torch.zeros(n_out)
    return p


def init_gnn_layer(gen, f_in: int, out_dim: int, n_heads: int, edge_dim: int = 4):
    """GraphTransformer (dgppo/nn/gnn.py:78-117): Dense_0..4 = q, k, v, edge(no bias), update."""
    hd = out_dim * n_heads
    return {
        "Dense_0": _dense(gen, f_in, hd), "Dense_1": _dense(gen, f_in, hd), "Dense_2": _dense(gen, f_in, hd),
        "Dense_3": _dense(gen, edge_dim, hd, bias=False), "Dense_4": _dense(gen, f_in, out_dim),
    }


def init_gnn(gen, node_dim: int, n_layers: int, msg_dim=32, out_dim=64, n_heads=3):
    p = {}
    f = node_dim
    for i in range(n_layers):
        d = out_dim if i == n_layers - 1 else msg_dim
        p[f"GraphTransformer_{i}"] = init_gnn_layer(gen, f, d, n_heads)
        f = d
    return p


def init_mlp(gen, f_in: int, hid=(64, 64)):
    p = {}
    f = f_in
    for i, h in enumerate(hid):
        p[f"Dense_{i}"] = _dense(gen, f, h)
        p[f"LayerNorm_{i}"] = {"scale": torch.ones(h), "bias": torch.zeros(h)}
        f = h
    return p


def init_gru(gen, f_in=64, hid=64):
    return {
        "ir": {"kernel": lecun_normal(gen, f_in, hid), "bias": torch.zeros(hid)},
        "iz": {"kernel": lecun_normal(gen, f_in, hid), "bias": torch.zeros(hid)},
        "in": {"kernel": lecun_normal(gen, f_in, hid), "bias": torch.zeros(hid)},
        "hr": {"kernel": orthogonal(gen, hid, hid)},
        "hz": {"kernel": orthogonal(gen, hid, hid)},
        "hn": {"kernel": orthogonal(gen, hid, hid), "bias": torch.zeros(hid)},
    }


def _init_rnn(gen, rnn_layers: int, lstm: bool = False):
    if lstm:
        return {f"LSTMCell_{3 * l + 2}": init_lstm(gen) for l in range(rnn_layers)}
    return {f"GRUCell_{2 * l + 1}": init_gru(gen) for l in range(rnn_layers)}


def init_policy(seed: int, node_dim: int, action_dim: int = 2, gnn_layers: int = 2, rnn_layers: int = 1, lstm: bool = False):
    """actor.pkl tree of SURVEY A.9 (dgppo/algo/module/policy.py:20-78,149-180).  rnn_layers = 0: --no-rnn."""
    gen = torch.Generator().manual_seed(seed)
    base = {"GraphTransformerGNN_0": init_gnn(gen, node_dim, gnn_layers), "PolicyGNNHead": init_mlp(gen, 64)}
    if rnn_layers > 0:
        base["RNN_0"] = _init_rnn(gen, rnn_layers, lstm)
    return {"params": {
        "PolicyNet_0": base,
        "ScaleHid": _dense(gen, 64, 64, scale=0.01),
        "OutputDenseMean": _dense(gen, 64, action_dim),
        "OutputDenseStdTrans": _dense(gen, 64, action_dim),
    }}


def init_value(seed: int, node_dim: int, n_out: int, gnn_layers: int, global_info: bool = False, rnn_layers: int = 1,
               lstm: bool = False):
    """Vl.pkl / Vh.pkl trees (dgppo/algo/module/value.py:15-79).  global_info: DecRStateFn(use_global_info=True), whose
    head sees [x_i | mean_j x_j] (value.py:66-68), i.e. a 128-wide first Dense."""
    gen = torch.Generator().manual_seed(seed)
    return {"params": {
        "GraphTransformerGNN_0": init_gnn(gen, node_dim, gnn_layers),
        "ValueGNNHead": init_mlp(gen, 128 if global_info else 64),
        **({"RNN_0": _init_rnn(gen, rnn_layers, lstm)} if rnn_layers > 0 else {}),
        "Dense_0": _dense(gen, 64, n_out),
    }}


def tree_map(fn, tree):
    if isinstance(tree, dict):
        return {k: tree_map(fn, v) for k, v in tree.items()}
    return fn(tree)


def tree_leaves(tree, prefix=""):
    """deterministic (sorted-key) traversal -> [(path, tensor)]."""
    out = []
    if isinstance(tree, dict):
        for k in sorted(tree.keys()):
            out += tree_leaves(tree[k], prefix + "/" + k)
    else:
        out.append((prefix, tree))
    return out


# ----------------------------------------------------------------------------------------------------------------------
# layers
# ----------------------------------------------------------------------------------------------------------------------
def dense(p, x):
    y = x @ p["kernel"]
    if "bias" in p:
        y = y + p["bias"]
    return y


def layer_norm(p, x, eps=1e-6):
    """flax nn.LayerNorm (use_fast_variance): var = max(E[x^2] - E[x]^2, 0)  [upstream]."""
    mean = x.mean(-1, keepdim=True)
    mean2 = (x * x).mean(-1, keepdim=True)
    var = torch.clamp(mean2 - mean * mean, min=0.0)
    return (x - mean) * torch.rsqrt(var + eps) * p["scale"] + p["bias"]


def mlp(p, x):
    """dgppo/nn/mlp.py:14-30 with hid=(64,64), act_final=True, use_layernorm=True."""
    i = 0
    while f"Dense_{i}" in p:
        x = F.relu(layer_norm(p[f"LayerNorm_{i}"], dense(p[f"Dense_{i}"], x)))
        i += 1
    return x


def gru_cell(p, h, x):
    """flax nn.GRUCell [upstream]: returns new_h."""
    r = torch.sigmoid(dense(p["ir"], x) + dense(p["hr"], h))
    z = torch.sigmoid(dense(p["iz"], x) + dense(p["hz"], h))
    n = torch.tanh(dense(p["in"], x) + r * dense(p["hn"], h))
    return (1.0 - z) * n + z * h


def init_lstm(gen, f_in=64, hid=64):
    """flax nn.LSTMCell(features=64): input Denses ii/if/ig/io without bias (lecun-normal), hidden Denses hi/hf/hg/ho with
    bias (orthogonal)  [upstream]"""
    p = {}
    for g_ in "ifgo":
        p["i" + g_] = {"kernel": lecun_normal(gen, f_in, hid)}
        p["h" + g_] = {"kernel": orthogonal(gen, hid, hid), "bias": torch.zeros(hid)}
    return p


def lstm_cell(p, c, h, x):
    """flax nn.LSTMCell [upstream]: returns (new_c, new_h)."""
    i = torch.sigmoid(dense(p["ii"], x) + dense(p["hi"], h))
    f = torch.sigmoid(dense(p["if"], x) + dense(p["hf"], h))
    g = torch.tanh(dense(p["ig"], x) + dense(p["hg"], h))
    o = torch.sigmoid(dense(p["io"], x) + dense(p["ho"], h))
    new_c = f * c + i * g
    return new_c, o * torch.tanh(new_c)


def rnn_apply(p_rnn, h, x):
    """RNN (dgppo/nn/rnn.py:14-30) with GRU cells.  p_rnn = the params of 'RNN_0' ({'GRUCell_{2l+1}': ...}: every layer
    instantiates the cell class once for the isinstance probe and once for use) or None (--no-rnn: the net
    has no cell and the carry passes through, policy.py:29-33).  h [..., L*64] packed carry -> (output, new packed carry)."""
    if p_rnn is None:
        return x, h
    names = sorted(p_rnn.keys(), key=lambda k: int(k.split("_")[1]))
    hs = []
    for l, name in enumerate(names):
        if name.startswith("LSTMCell"):   # packed carry [c_l | h_l] per layer (rnn.py:23-24: carries 0 and 1 of the layer)
            cl, hl = lstm_cell(p_rnn[name], h[..., (2 * l) * 64:(2 * l + 1) * 64], h[..., (2 * l + 1) * 64:(2 * l + 2) * 64], x)
            hs += [cl, hl]
        else:
            hl = gru_cell(p_rnn[name], h[..., l * 64:(l + 1) * 64], x)
            hs.append(hl)
        x = hl
    return x, torch.cat(hs, dim=-1)


def carry_width(tree) -> int:
    """width of the packed carry of a policy / value tree: 64 per stacked cell (64 of pass-through zeros without a cell)"""
    p = tree["params"]
    p = p.get("PolicyNet_0", p)
    cells = p.get("RNN_0", {})
    return 64 * max(len(cells), 1) * (2 if any(k.startswith("LSTMCell") for k in cells) else 1)


def segment_softmax(logits: Tensor, seg: Tensor, num_segments: int) -> Tensor:
    """jraph.segment_softmax [upstream]: per segment subtract max, exp, divide by sum. logits [E, H]."""
    H = logits.shape[1]
    idx = seg[:, None].expand(-1, H)
    mx = torch.full((num_segments, H), -torch.inf, dtype=logits.dtype).scatter_reduce(0, idx, logits, "amax", include_self=True)
    ex = torch.exp(logits - mx[seg].detach())
    den = torch.zeros(num_segments, H, dtype=logits.dtype).index_add(0, seg, ex)
    return ex / den[seg]


def gnn_layer(p, nodes, edges, senders, receivers, n_heads: int, out_dim: int):
    """GraphTransformer via GNNUpdate (dgppo/nn/gnn.py:27-41,85-117).  nodes [N,F], edges [E,4], int64 indices."""
    N = nodes.shape[0]
    xs = nodes[senders]
    xr = nodes[receivers]
    q = dense(p["Dense_0"], xr).reshape(-1, n_heads, out_dim)
    k = dense(p["Dense_1"], xs).reshape(-1, n_heads, out_dim)
    v = dense(p["Dense_2"], xs).reshape(-1, n_heads, out_dim)
    e = dense(p["Dense_3"], edges).reshape(-1, n_heads, out_dim)
    attn = (q * k).sum(-1) / math.sqrt(out_dim)
    attn = segment_softmax(attn, receivers, N)[:, :, None]
    msgs = (attn * (v + e)).mean(dim=1)
    aggr = torch.zeros(N, out_dim, dtype=nodes.dtype).index_add(0, receivers, msgs)
    return F.relu(dense(p["Dense_4"], nodes) + aggr)


def gnn(p, graph: Dict[str, Tensor], n_agents: int, n_heads=3, msg_dim=32, out_dim=64):
    """GraphTransformerGNN (dgppo/nn/gnn.py:127-142) on a BATCH of equally-shaped graphs, then type_nodes(0, n)
    (utils/graph.py:115-127 == rows [0, n)).  graph tensors have a leading batch axis G."""
    nodes, edges = graph["nodes"], graph["edges"]
    G, N, _ = nodes.shape
    off = (torch.arange(G) * N)[:, None]
    senders = (graph["senders"].long() + off).reshape(-1)
    receivers = (graph["receivers"].long() + off).reshape(-1)
    x = nodes.reshape(G * N, -1)
    ef = edges.reshape(-1, edges.shape[-1])
    n_layers = len(p)
    for i in range(n_layers):
        d = out_dim if i == n_layers - 1 else msg_dim
        x = gnn_layer(p[f"GraphTransformer_{i}"], x, ef, senders, receivers, n_heads, d)
    return x.reshape(G, N, -1)[:, :n_agents]


# ----------------------------------------------------------------------------------------------------------------------
# actor  (dgppo/algo/module/policy.py:20-78,185-212 ; distribution.py:10-46)
# ----------------------------------------------------------------------------------------------------------------------
STD_INIT_INV = math.log(math.exp(0.5) - 1.0)
STD_MIN = 1e-5
THRESH = 0.999
INV_THRESH = math.atanh(THRESH)
LOG_EPS = math.log(1.0 - THRESH)


def policy_net(pp, graph, h, n_agents):
    """PolicyNet: GNN -> agents -> MLP -> GRU.  h [G, n, 64] (rnn_layers = 1, carry axis squeezed) -> (feat, new_h)."""
    p = pp["params"]["PolicyNet_0"]
    x = gnn(p["GraphTransformerGNN_0"], graph, n_agents)
    x = mlp(p["PolicyGNNHead"], x)
    return rnn_apply(p.get("RNN_0"), h, x)


def policy_dist(pp, graph, h, n_agents):
    x, new_h = policy_net(pp, graph, h, n_agents)
    p = pp["params"]
    u = dense(p["ScaleHid"], x)
    mean = dense(p["OutputDenseMean"], u)
    std = F.softplus(dense(p["OutputDenseStdTrans"], u) + STD_INIT_INV) + STD_MIN
    return mean, std, new_h


def tanh_fldj(x):
    """tfb.Tanh forward_log_det_jacobian [upstream]: 2*(log 2 - x - softplus(-2x))."""
    return 2.0 * (math.log(2.0) - x - F.softplus(-2.0 * x))


def normal_log_prob(x, mean, std):
    z = (x - mean) / std
    return -0.5 * z * z - torch.log(std) - 0.5 * LOG_2PI


def tanh_normal_log_prob(action, mean, std):
    """TanhTransformedDistribution.log_prob (distribution.py:25-35), summed over the action dim (Independent)."""
    a = torch.clamp(action, -THRESH, THRESH)
    x = torch.atanh(a)
    lp = normal_log_prob(x, mean, std) - tanh_fldj(x)
    left = torch.special.log_ndtr((-INV_THRESH - mean) / std) - LOG_EPS
    right = torch.special.log_ndtr(-(INV_THRESH - mean) / std) - LOG_EPS
    lp = torch.where(a <= -THRESH, left, torch.where(a >= THRESH, right, lp))
    return lp.sum(-1)


def tanh_normal_entropy(mean, std, eps_hat):
    """distribution.py:37-43: Normal entropy + fldj at ONE sample drawn with a trace-time-constant key:
    eps_hat is that fixed [n, action_dim] standard-normal matrix."""
    y = mean + std * eps_hat
    ent = 0.5 + 0.5 * LOG_2PI + torch.log(std) + tanh_fldj(y)
    return ent.sum(-1)


def policy_sample(pp, graph, h, n_agents, eps):
    """PPOPolicy.sample_action (policy.py:196-203) with injected noise eps [G,n,2]."""
    mean, std, new_h = policy_dist(pp, graph, h, n_agents)
    action = torch.tanh(mean + std * eps)
    return action, tanh_normal_log_prob(action, mean, std), new_h


def policy_mode(pp, graph, h, n_agents):
    """PPOPolicy.get_action (policy.py:191-194): tanh(mean)."""
    mean, std, new_h = policy_dist(pp, graph, h, n_agents)
    return torch.tanh(mean), new_h


def policy_eval(pp, graph, action, h, n_agents, eps_hat):
    """PPOPolicy.eval_action (policy.py:205-212)."""
    mean, std, new_h = policy_dist(pp, graph, h, n_agents)
    return tanh_normal_log_prob(action, mean, std), tanh_normal_entropy(mean, std, eps_hat), new_h


# ----------------------------------------------------------------------------------------------------------------------
# value nets  (dgppo/algo/module/value.py:15-79)
# ----------------------------------------------------------------------------------------------------------------------
def value_Vl(vp, graph, h, n_agents):
    """RStateFn: GNN -> agents -> mean -> MLP -> GRU -> Dense 1.  h [G,1,64] -> (Vl [G], new_h [G,1,64])."""
    p = vp["params"]
    x = gnn(p["GraphTransformerGNN_0"], graph, n_agents)
    x = x.mean(dim=1, keepdim=True)
    x = mlp(p["ValueGNNHead"], x)
    x, new_h = rnn_apply(p.get("RNN_0"), h, x)
    return dense(p["Dense_0"], x)[:, 0, 0], new_h


def value_Vh(vp, graph, h, n_agents, global_info: bool = False):
    """DecRStateFn: GNN -> agents [-> concat the mean over agents, tiled (use_global_info, value.py:66-68)] -> MLP -> GRU ->
    Dense n_cost.  h [G,n,64] -> [G,n,n_cost]."""
    p = vp["params"]
    x = gnn(p["GraphTransformerGNN_0"], graph, n_agents)
    if global_info:
        x = torch.cat([x, x.mean(dim=1, keepdim=True).expand(-1, n_agents, -1)], dim=-1)
    x = mlp(p["ValueGNNHead"], x)
    x, new_h = rnn_apply(p.get("RNN_0"), h, x)
    return dense(p["Dense_0"], x), new_h


def graph_to_torch(g: Dict[str, np.ndarray]) -> Dict[str, Tensor]:
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in g.items()}
