"""Pins oracle/nn_torch.py (the NN half of the oracle) against INDEPENDENT implementations importable in this
container — the reference itself cannot run here (no jax / flax / jraph / tfp: SURVEY F3, §8c), so these are the
strongest pins available and the oracle stays "parity unpinned" with respect to outputs of the reference.

 * GRU cell            vs torch.nn.GRUCell with the flax -> torch gate mapping   (dgppo/nn/rnn.py:10-30, flax GRUCell)
 * LayerNorm           vs torch.nn.functional.layer_norm (eps = 1e-6)            (dgppo/nn/mlp.py:17-29)
 * tanh-Normal log-prob vs torch.distributions (Normal + TanhTransform) in the interior and scipy.stats.norm.logcdf on
                       the +-0.999 branches                                      (dgppo/algo/module/distribution.py:17-35)
 * segment softmax / GraphTransformer layer: hand-set weights with a closed-form answer (dgppo/nn/gnn.py:78-117)
 * SURVEY §8(c)(3) properties (hypothesis): agent-permutation equivariance of the actor and Vh, invariance of Vl,
   masked edge => zero influence, the pad node never reaches an agent
 * autograd of the twin vs finite differences (float64 gradcheck)
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F
from hypothesis import given, settings, strategies as st
from scipy.stats import norm as sp_norm

from oracle import env_np as E
from oracle import nn_torch as T


def _gen(seed):
    return torch.Generator().manual_seed(seed)


# ----------------------------------------------------------------------------------------------------------------------
# GRU
# ----------------------------------------------------------------------------------------------------------------------
def _torch_gru_from_flax(p, f_in, hid):
    """torch.nn.GRUCell: weight_ih [3H, in] in (r, z, n) order, y = W x + b; flax kernels are [in, out]."""
    cell = torch.nn.GRUCell(f_in, hid)
    with torch.no_grad():
        cell.weight_ih.copy_(torch.cat([p["ir"]["kernel"].T, p["iz"]["kernel"].T, p["in"]["kernel"].T], 0))
        cell.weight_hh.copy_(torch.cat([p["hr"]["kernel"].T, p["hz"]["kernel"].T, p["hn"]["kernel"].T], 0))
        cell.bias_ih.copy_(torch.cat([p["ir"]["bias"], p["iz"]["bias"], p["in"]["bias"]]))
        # flax GRUCell has no recurrent bias on r and z; the n-gate's recurrent bias sits INSIDE r * (W_hn h + b_hn)
        cell.bias_hh.copy_(torch.cat([torch.zeros(hid), torch.zeros(hid), p["hn"]["bias"]]))
    return cell


@pytest.mark.parametrize("f_in,hid,rows", [(64, 64, 37), (5, 3, 4)])
def test_gru_cell_matches_torch_grucell(f_in, hid, rows):
    g = _gen(3)
    p = T.init_gru(g, f_in, hid)
    for k in ("ir", "iz", "in", "hn"):   # non-zero biases so the bias placement is exercised
        p[k]["bias"] = 0.3 * torch.randn(hid, generator=g)
    x = torch.randn(rows, f_in, generator=g)
    h = torch.randn(rows, hid, generator=g)
    want = _torch_gru_from_flax(p, f_in, hid)(x, h)
    got = T.gru_cell(p, h, x)
    assert torch.allclose(got, want, atol=1e-6), float((got - want).abs().max())


def test_gru_cell_hand_weights():
    """1-d cell with hand-set weights: r = s(1*x + 0.5*h), z = s(-x + 0.25 h + 0.1), n = tanh(2x + 0.2 + r*(h - 0.3))."""
    one = lambda v: torch.tensor([[v]], dtype=torch.float32)
    p = {"ir": {"kernel": one(1.0), "bias": torch.zeros(1)}, "iz": {"kernel": one(-1.0), "bias": torch.tensor([0.1])},
         "in": {"kernel": one(2.0), "bias": torch.tensor([0.2])}, "hr": {"kernel": one(0.5)}, "hz": {"kernel": one(0.25)},
         "hn": {"kernel": one(1.0), "bias": torch.tensor([-0.3])}}
    x, h = 0.7, -0.4
    s = lambda v: 1.0 / (1.0 + math.exp(-v))
    r = s(x + 0.5 * h)
    z = s(-x + 0.25 * h + 0.1)
    n = math.tanh(2 * x + 0.2 + r * (h - 0.3))
    want = (1 - z) * n + z * h
    got = float(T.gru_cell(p, one(h), one(x)))
    assert abs(got - want) < 1e-6


# ----------------------------------------------------------------------------------------------------------------------
# LayerNorm / MLP
# ----------------------------------------------------------------------------------------------------------------------
def test_layer_norm_matches_torch():
    g = _gen(5)
    x = 3.0 * torch.randn(50, 64, generator=g) + 1.5
    p = {"scale": torch.randn(64, generator=g), "bias": torch.randn(64, generator=g)}
    want = F.layer_norm(x, (64,), p["scale"], p["bias"], eps=1e-6)
    got = T.layer_norm(p, x)
    assert torch.allclose(got, want, atol=2e-6), float((got - want).abs().max())


def test_layer_norm_hand_values():
    """x = [1, 2, 3, 6]: mean 3, var 3.5 -> (x - 3)/sqrt(3.5 + 1e-6) * 2 + 1."""
    x = torch.tensor([[1.0, 2.0, 3.0, 6.0]])
    got = T.layer_norm({"scale": torch.full((4,), 2.0), "bias": torch.ones(4)}, x)
    want = (np.array([1.0, 2.0, 3.0, 6.0]) - 3.0) / math.sqrt(3.5 + 1e-6) * 2.0 + 1.0
    np.testing.assert_allclose(got.numpy()[0], want, atol=1e-6)


def test_layer_norm_constant_row_is_bias():
    """variance clamps at zero (use_fast_variance): a constant row maps to the bias, never NaN."""
    got = T.layer_norm({"scale": torch.ones(8), "bias": torch.full((8,), 0.25)}, torch.full((2, 8), 1e3))
    assert torch.isfinite(got).all() and torch.allclose(got, torch.full((2, 8), 0.25), atol=1e-3)


def test_mlp_is_dense_ln_relu_twice():
    g = _gen(6)
    p = T.init_mlp(g, 64)
    x = torch.randn(9, 64, generator=g)
    y = x
    for i in range(2):
        d = p[f"Dense_{i}"]
        y = F.relu(F.layer_norm(y @ d["kernel"] + d["bias"], (64,), p[f"LayerNorm_{i}"]["scale"], p[f"LayerNorm_{i}"]["bias"], 1e-6))
    assert torch.allclose(T.mlp(p, x), y, atol=2e-6)


# ----------------------------------------------------------------------------------------------------------------------
# tanh-Normal
# ----------------------------------------------------------------------------------------------------------------------
def test_tanh_normal_log_prob_interior_matches_torch_distributions():
    g = _gen(7)
    mean = torch.randn(200, 2, generator=g)
    std = F.softplus(torch.randn(200, 2, generator=g)) + 1e-3
    x = mean + std * torch.randn(200, 2, generator=g)
    a = torch.tanh(x).clamp(-0.99, 0.99)                     # strictly inside the +-0.999 clip
    base = torch.distributions.Normal(mean.double(), std.double())
    dist = torch.distributions.TransformedDistribution(base, [torch.distributions.transforms.TanhTransform(cache_size=0)])
    want = dist.log_prob(a.double()).sum(-1)
    got = T.tanh_normal_log_prob(a, mean, std)
    assert torch.allclose(got.double(), want, atol=5e-5, rtol=1e-5), float((got.double() - want).abs().max())


def test_tanh_normal_log_prob_edge_branches_match_scipy():
    """|a| >= 0.999: log CDF mass beyond the threshold minus log(1 - 0.999) (distribution.py:28-33)."""
    mean = torch.tensor([[0.3, -1.2], [2.5, 0.1], [-4.0, 5.0]])
    std = torch.tensor([[0.5, 1.5], [0.2, 0.9], [1.0, 2.0]])
    thr = math.atanh(0.999)
    log_eps = math.log(1.0 - 0.999)
    for sign in (-1.0, 1.0):
        a = torch.full_like(mean, sign)                      # tanh saturated to exactly +-1 -> clipped to +-0.999
        got = T.tanh_normal_log_prob(a, mean, std).double().numpy()
        m, s = mean.double().numpy(), std.double().numpy()
        if sign < 0:
            want = sp_norm.logcdf((-thr - m) / s) - log_eps
        else:
            want = sp_norm.logcdf(-(thr - m) / s) - log_eps
        np.testing.assert_allclose(got, want.sum(-1), rtol=2e-6, atol=2e-6)
    # mixed row: one dim interior, one dim at the upper edge
    a = torch.tensor([[0.25, 1.0]])
    m1, s1 = torch.tensor([[0.1, 0.4]]), torch.tensor([[0.7, 0.3]])
    x0 = math.atanh(0.25)
    interior = sp_norm.logpdf(x0, 0.1, 0.7) - math.log(1.0 - 0.25 ** 2)
    edge = sp_norm.logcdf(-(thr - 0.4) / 0.3) - log_eps
    assert abs(float(T.tanh_normal_log_prob(a, m1, s1)) - (interior + edge)) < 1e-5


def test_tanh_fldj_is_log_one_minus_tanh_squared():
    x = torch.linspace(-3, 3, 41, dtype=torch.float64)
    np.testing.assert_allclose(T.tanh_fldj(x).numpy(), np.log(1.0 - np.tanh(x.numpy()) ** 2), atol=1e-12)


def test_entropy_known_value():
    """distribution.py:37-43 with eps_hat = 0: H = sum_d [0.5 ln(2 pi e) + ln s + ln(1 - tanh(m)^2)]."""
    mean, std = torch.tensor([[0.2, -0.5]]), torch.tensor([[0.5, 2.0]])
    want = sum(0.5 * math.log(2 * math.pi * math.e) + math.log(s) + math.log(1 - math.tanh(m) ** 2)
               for m, s in ((0.2, 0.5), (-0.5, 2.0)))
    assert abs(float(T.tanh_normal_entropy(mean, std, torch.zeros(1, 2))) - want) < 1e-6


def test_std_parametrisation():
    """policy.py:70-72: softplus(0 + ln(e^0.5 - 1)) + 1e-5 = 0.5 + 1e-5 at zero pre-activation."""
    assert abs(float(F.softplus(torch.tensor(T.STD_INIT_INV))) - 0.5) < 1e-7
    assert abs(T.STD_INIT_INV - (-0.432752)) < 1e-6 and abs(T.INV_THRESH - 3.8002012) < 1e-6 and abs(T.LOG_EPS + 6.9077553) < 1e-6


# ----------------------------------------------------------------------------------------------------------------------
# segment softmax / one GraphTransformer layer with hand-set weights
# ----------------------------------------------------------------------------------------------------------------------
def test_segment_softmax_hand_case():
    logits = torch.tensor([[0.0], [math.log(3.0)], [5.0], [1.0], [1.0]])
    seg = torch.tensor([0, 0, 1, 2, 2])
    got = T.segment_softmax(logits, seg, 4)[:, 0].numpy()       # segment 3 is empty
    np.testing.assert_allclose(got, [0.25, 0.75, 1.0, 0.5, 0.5], atol=1e-7)


def test_segment_softmax_matches_dense_softmax_per_segment():
    g = _gen(9)
    logits = 4.0 * torch.randn(30, 3, generator=g)
    seg = torch.randint(0, 5, (30,), generator=g)
    got = T.segment_softmax(logits, seg, 5)
    for s in range(5):
        m = seg == s
        if m.any():
            assert torch.allclose(got[m], torch.softmax(logits[m], 0), atol=1e-6)


def test_gnn_layer_hand_weights():
    """F = 1 feature, D = 1, H = 1 head, identity-like weights: q = x_r, k = x_s, v = x_s, e = 2*edge, update = x.
    Receiver 0 has senders 1 and 2: logits x0*x1, x0*x2 -> softmax -> m = a1*(x1 + 2 e1) + a2*(x2 + 2 e2);
    x0' = relu(x0 + m); nodes without incoming edges get relu(x)."""
    one = lambda v: torch.tensor([[v]], dtype=torch.float32)
    p = {"Dense_0": {"kernel": one(1.0), "bias": torch.zeros(1)}, "Dense_1": {"kernel": one(1.0), "bias": torch.zeros(1)},
         "Dense_2": {"kernel": one(1.0), "bias": torch.zeros(1)}, "Dense_3": {"kernel": one(2.0)},
         "Dense_4": {"kernel": one(1.0), "bias": torch.zeros(1)}}
    nodes = torch.tensor([[1.0], [0.5], [-2.0]])
    edges = torch.tensor([[0.1], [0.3]])
    senders, receivers = torch.tensor([1, 2]), torch.tensor([0, 0])
    got = T.gnn_layer(p, nodes, edges, senders, receivers, 1, 1)[:, 0].numpy()
    l1, l2 = 1.0 * 0.5, 1.0 * -2.0
    a1 = math.exp(l1) / (math.exp(l1) + math.exp(l2))
    a2 = 1.0 - a1
    m = a1 * (0.5 + 0.2) + a2 * (-2.0 + 0.6)
    np.testing.assert_allclose(got, [max(1.0 + m, 0.0), 0.5, 0.0], atol=1e-6)


def test_gnn_layer_head_mean_and_scale():
    """two heads, D = 2: logits are <q_h, k_h>/sqrt(D) per head and messages are averaged over heads (gnn.py:100-107)."""
    g = _gen(10)
    H, D, Fin = 2, 2, 3
    p = T.init_gnn_layer(g, Fin, D, H, edge_dim=4)
    for k in ("Dense_0", "Dense_1", "Dense_2", "Dense_4"):
        p[k]["bias"] = 0.1 * torch.randn_like(p[k]["bias"])
    nodes = torch.randn(4, Fin, generator=g)
    edges = torch.randn(3, 4, generator=g)
    senders, receivers = torch.tensor([1, 2, 3]), torch.tensor([0, 0, 0])
    got = T.gnn_layer(p, nodes, edges, senders, receivers, H, D)
    q = (nodes[0] @ p["Dense_0"]["kernel"] + p["Dense_0"]["bias"]).view(H, D)
    k = (nodes[1:] @ p["Dense_1"]["kernel"] + p["Dense_1"]["bias"]).view(3, H, D)
    v = (nodes[1:] @ p["Dense_2"]["kernel"] + p["Dense_2"]["bias"]).view(3, H, D)
    e = (edges @ p["Dense_3"]["kernel"]).view(3, H, D)
    att = torch.softmax((q[None] * k).sum(-1) / math.sqrt(D), 0)                # [3, H]
    m = (att[:, :, None] * (v + e)).mean(1).sum(0)
    want0 = F.relu(nodes[0] @ p["Dense_4"]["kernel"] + p["Dense_4"]["bias"] + m)
    assert torch.allclose(got[0], want0, atol=1e-6)
    assert torch.allclose(got[1:], F.relu(nodes[1:] @ p["Dense_4"]["kernel"] + p["Dense_4"]["bias"]), atol=1e-6)


# ----------------------------------------------------------------------------------------------------------------------
# properties
# ----------------------------------------------------------------------------------------------------------------------
def _scene(kind_name, n, n_obs, seed):
    """random but valid scene + its graph; close agents so that the agent-agent mask has both values."""
    rng = np.random.default_rng(seed)
    cfg = E.EnvCfg(E.KIND_NAMES[kind_name], n_agents=n, n_obs=n_obs)
    sd = cfg.state_dim
    agent = np.zeros((1, n, sd), np.float32)
    agent[..., :2] = rng.uniform(0.3, 1.2, (1, n, 2))
    agent[..., 2:] = rng.uniform(-0.4, 0.4, (1, n, sd - 2))
    goal = np.zeros((1, n, sd), np.float32)
    goal[..., :2] = rng.uniform(0, 1.5, (1, n, 2))
    obst, hits = None, None
    if cfg.is_lidar:
        # hit points: some close to the agent (active edges), some far misses (masked edges, +-5e5 like the reference)
        hits = agent[:, :, None, :2] + rng.uniform(-0.3, 0.3, (1, n, cfg.top_k, 2)).astype(np.float32)
        miss = rng.random((1, n, cfg.top_k)) < 0.4
        hits[miss] = hits[miss] + np.float32(5e5)
        hits = hits.astype(np.float32)
        obst = np.zeros((1, n_obs, 16), np.float32)
    elif n_obs > 0:
        obst = np.zeros((1, n_obs, sd), np.float32)
        obst[..., :2] = rng.uniform(0, 1.5, (1, n_obs, 2))
    return cfg, agent, goal, obst, hits


def _graph(cfg, agent, goal, obst, hits):
    return T.graph_to_torch(E.get_graph(cfg, agent, goal, obst, hits))


def _nets(cfg, seed):
    pol = T.init_policy(seed, cfg.node_dim, 2, 2)
    vl = T.init_value(seed + 1, cfg.node_dim, 1, 2)
    vh = T.init_value(seed + 2, cfg.node_dim, 2, 1)
    pol["params"]["ScaleHid"]["kernel"] = T.orthogonal(_gen(seed), 64, 64, 0.5)   # visible action means
    return pol, vl, vh


@settings(max_examples=12, deadline=None)
@given(seed=st.integers(0, 10_000), kind=st.sampled_from(["LidarSpread", "LidarTarget", "MPESpread"]), n=st.integers(2, 5))
def test_agent_permutation_equivariance_and_invariance(seed, kind, n):
    """Relabelling the agents (with their goals / hit points, as the env would) permutes the actor's and Vh's per-agent
    outputs and leaves Vl unchanged (the GNN is a function of the graph, not of the node numbering)."""
    cfg, agent, goal, obst, hits = _scene(kind, n, 2, seed)
    perm = np.random.default_rng(seed + 1).permutation(n)
    g0 = _graph(cfg, agent, goal, obst, hits)
    g1 = _graph(cfg, agent[:, perm], goal[:, perm], obst, None if hits is None else hits[:, perm])
    pol, vl, vh = _nets(cfg, seed % 7)
    h = 0.1 * torch.randn(1, n, 64, generator=_gen(seed))
    hp = h[:, torch.from_numpy(perm)]
    with torch.no_grad():
        a0, _ = T.policy_mode(pol, g0, h, n)
        a1, _ = T.policy_mode(pol, g1, hp, n)
        v0, _ = T.value_Vh(vh, g0, h, n)
        v1, _ = T.value_Vh(vh, g1, hp, n)
        hl = 0.1 * torch.randn(1, 1, 64, generator=_gen(seed + 3))
        l0, _ = T.value_Vl(vl, g0, hl, n)
        l1, _ = T.value_Vl(vl, g1, hl, n)
    assert torch.allclose(a0[:, torch.from_numpy(perm)], a1, atol=2e-6)
    assert torch.allclose(v0[:, torch.from_numpy(perm)], v1, atol=2e-5)
    assert torch.allclose(l0, l1, atol=2e-5)


@settings(max_examples=12, deadline=None)
@given(seed=st.integers(0, 10_000), kind=st.sampled_from(["LidarSpread", "MPESpread"]))
def test_masked_edge_has_zero_influence_and_pad_never_reaches_agents(seed, kind):
    """(1) Changing the FEATURES of a masked edge (receiver = sender = pad) or the feature row of a node that only
    masked edges point to leaves every agent output bit-identical.  (2) Writing garbage into the pad node's feature row
    leaves every agent output bit-identical (pad only ever sends to pad)."""
    n = 4
    cfg, agent, goal, obst, hits = _scene(kind, n, 2, seed)
    g0 = _graph(cfg, agent, goal, obst, hits)
    N = cfg.num_nodes
    pad = N - 1
    masked = (g0["receivers"][0] == pad)
    assert bool((g0["senders"][0][masked] == pad).all())         # masked endpoints are BOTH re-routed to the pad node
    pol, vl, vh = _nets(cfg, 3)
    h = 0.1 * torch.randn(1, n, 64, generator=_gen(seed))
    hl = torch.zeros(1, 1, 64)

    def outs(g):
        with torch.no_grad():
            return T.policy_mode(pol, g, h, n)[0], T.value_Vh(vh, g, h, n)[0], T.value_Vl(vl, g, hl, n)[0]
    base = outs(g0)
    if bool(masked.any()):
        g1 = {k: v.clone() for k, v in g0.items()}
        g1["edges"][0][masked] += 123.0
        for x, y in zip(base, outs(g1)):
            assert torch.equal(x, y)
    g2 = {k: v.clone() for k, v in g0.items()}
    g2["nodes"][0, pad] = torch.tensor(np.random.default_rng(seed).normal(size=cfg.node_dim).astype(np.float32)) * 50.0
    for x, y in zip(base, outs(g2)):
        assert torch.equal(x, y)
    # index contract of to_padded (utils/graph.py:212-247): every endpoint in [0, N-1]
    for key in ("senders", "receivers"):
        assert int(g0[key].min()) >= 0 and int(g0[key].max()) <= pad


def test_far_lidar_hit_node_cannot_influence_its_agent():
    """a LiDAR miss (hit point 5e5 away) is masked: moving it further changes nothing."""
    cfg, agent, goal, obst, hits = _scene("LidarSpread", 3, 2, 5)
    hits[0, 1, 2] = agent[0, 1, :2] + np.float32(5e5)
    g0 = _graph(cfg, agent, goal, obst, hits)
    hits2 = hits.copy()
    hits2[0, 1, 2] = agent[0, 1, :2] - np.float32(3e5)
    g1 = _graph(cfg, agent, goal, obst, hits2)
    pol, _, vh = _nets(cfg, 1)
    h = torch.zeros(1, 3, 64)
    with torch.no_grad():
        assert torch.equal(T.policy_mode(pol, g0, h, 3)[0], T.policy_mode(pol, g1, h, 3)[0])
        assert torch.equal(T.value_Vh(vh, g0, h, 3)[0], T.value_Vh(vh, g1, h, 3)[0])


# ----------------------------------------------------------------------------------------------------------------------
# gradient twin vs finite differences
# ----------------------------------------------------------------------------------------------------------------------
def test_autograd_twin_matches_finite_differences():
    cfg, agent, goal, obst, hits = _scene("LidarSpread", 2, 1, 11)
    g = {k: (v.double() if v.dtype == torch.float32 else v) for k, v in _graph(cfg, agent, goal, obst, hits).items()}
    vh = T.tree_map(lambda t: t.double(), T.init_value(4, cfg.node_dim, 2, 1))
    leaves = T.tree_leaves(vh)
    names = [k for k, _ in leaves if "GraphTransformer_0/Dense_0/kernel" in k or "GRUCell_1/hn/kernel" in k or "LayerNorm_0/scale" in k]
    h = 0.1 * torch.randn(1, 2, 64, generator=_gen(2), dtype=torch.float64)

    def set_leaf(tree, path, val):
        keys = [k for k in path.split("/") if k]
        for k in keys[:-1]:
            tree = tree[k]
        tree[keys[-1]] = val

    for name in names:
        base = dict(leaves)[name]

        def f(w, name=name):
            tree = T.tree_map(lambda t: t, vh)
            set_leaf(tree, name, w)
            return T.value_Vh(tree, g, h, 2)[0]
        w = base.clone().requires_grad_()
        assert torch.autograd.gradcheck(f, (w,), eps=1e-6, atol=1e-5, rtol=1e-4, nondet_tol=1e-9), name


def test_lstm_cell_matches_torch_lstmcell():
    """oracle lstm_cell (flax nn.LSTMCell restated: input Denses without bias, hidden Denses with bias, gates i f g o,
    c' = f c + i g, h' = o tanh(c')) vs torch.nn.LSTMCell with the same gate order."""
    g = _gen(12)
    p = T.init_lstm(g, 64, 64)
    for k in ("hi", "hf", "hg", "ho"):
        p[k]["bias"] = 0.3 * torch.randn(64, generator=g)
    cell = torch.nn.LSTMCell(64, 64)
    with torch.no_grad():
        cell.weight_ih.copy_(torch.cat([p["i" + q]["kernel"].T for q in "ifgo"], 0))
        cell.weight_hh.copy_(torch.cat([p["h" + q]["kernel"].T for q in "ifgo"], 0))
        cell.bias_ih.zero_()
        cell.bias_hh.copy_(torch.cat([p["h" + q]["bias"] for q in "ifgo"]))
    x, h, c = (torch.randn(11, 64, generator=g) for _ in range(3))
    h_w, c_w = cell(x, (h, c))
    c_g, h_g = T.lstm_cell(p, c, h, x)
    assert torch.allclose(h_g, h_w, atol=1e-6) and torch.allclose(c_g, c_w, atol=1e-6)
    # stacked application through rnn_apply: packed carry [c_0 | h_0 | c_1 | h_1]
    prnn = {"LSTMCell_2": p, "LSTMCell_5": T.init_lstm(g, 64, 64)}
    carry = torch.randn(11, 256, generator=g)
    out, new = T.rnn_apply(prnn, carry, x)
    c0, h0 = T.lstm_cell(prnn["LSTMCell_2"], carry[:, :64], carry[:, 64:128], x)
    c1, h1 = T.lstm_cell(prnn["LSTMCell_5"], carry[:, 128:192], carry[:, 192:], h0)
    assert torch.equal(out, h1) and torch.equal(new, torch.cat([c0, h0, c1, h1], -1))
    assert T.carry_width({"params": {"RNN_0": prnn}}) == 256
