"""GPU parity of the network building blocks and of the three composed networks (HIP, via the C ABI) against the
torch-CPU oracle in the reference's per-edge form (oracle/nn_torch.py).  fp32 tolerance: 1e-5 on O(1) outputs
(north_star), gradients 2e-5 relative to the gradient's scale."""
import math

import numpy as np
import pytest
import torch

from oracle import env_np as E
from oracle import nn_torch as T

pytestmark = pytest.mark.gpu


def _close(got, want, tol=1e-5, name=""):
    got = got.detach().cpu().double()
    want = want.detach().cpu().double()
    assert got.shape == want.shape, (name, got.shape, want.shape)
    scale = max(1.0, float(want.abs().max()))
    err = float((got - want).abs().max())
    assert err <= tol * scale, f"{name}: max abs err {err:.3e} (scale {scale:.3e})"


@pytest.mark.parametrize("M,K,N,act,trans", [(1000, 64, 64, 1, False), (777, 7, 24, 0, False), (130, 48, 32, 1, False),
                                             (64, 144, 64, 0, False), (513, 64, 192, 0, False), (300, 64, 4, 0, False),
                                             (300, 4, 64, 0, True), (1000, 192, 64, 0, True), (257, 64, 141, 0, True)])
def test_dense_fwd(cuda, M, K, N, act, trans):
    from dgppo_amd import ops_nn as K_
    g = torch.Generator().manual_seed(M + K + N)
    X = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) if trans else torch.randn(K, N, generator=g)
    b = torch.randn(N, generator=g)
    Y0 = torch.randn(M, N, generator=g)
    want = X @ (W.T if trans else W) + b
    if act:
        want = torch.relu(want)
    Y = torch.empty(M, N, device=cuda)
    K_.dense_fwd(X.to(cuda), W.to(cuda), b.to(cuda), Y, act=act, trans_w=trans)
    _close(Y, want, 2e-6 * math.sqrt(K), "dense")
    # accumulate + strided views (column slices of wider matrices)
    Xw = torch.zeros(M, K + 5, device=cuda)
    Xw[:, 2:2 + K] = X.to(cuda)
    Yw = torch.zeros(M, N + 3, device=cuda)
    Yw[:, 1:1 + N] = Y0.to(cuda)
    K_.dense_fwd(Xw[:, 2:2 + K], W.to(cuda), None, Yw[:, 1:1 + N], accumulate=True, trans_w=trans)
    _close(Yw[:, 1:1 + N], Y0 + X @ (W.T if trans else W), 2e-6 * math.sqrt(K), "dense acc")
    assert float(Yw[:, 0].abs().max()) == 0 and float(Yw[:, 1 + N:].abs().max()) == 0


@pytest.mark.parametrize("M,K,N", [(5000, 64, 64), (999, 48, 32), (4096, 144, 64), (2000, 64, 192), (700, 8, 24),
                                   (1500, 64, 4), (100, 32, 96)])
def test_dense_bwd_w(cuda, M, K, N):
    from dgppo_amd import ops_nn as K_
    g = torch.Generator().manual_seed(M * 3 + K + N)
    X = torch.randn(M, K, generator=g)
    dY = torch.randn(M, N, generator=g)
    dW = torch.zeros(K, N, device=cuda)
    db = torch.zeros(N, device=cuda)
    K_.dense_bwd_w(X.to(cuda), dY.to(cuda), dW, db)
    _close(dW, X.double().T @ dY.double(), 3e-6 * math.sqrt(M), "dW")
    _close(db, dY.double().sum(0), 3e-6 * math.sqrt(M), "db")


def test_dense_kernels_random_shape_sweep(cuda):
    """seeded sweep over the instantiation space of the persistent dense kernels: K in [1, 256] (every padded-K variant,
    K % 4 != 0 takes the scalar staging path), N in [1, 192], ragged M incl. M < one tile, strided X / Y views, transposed
    W, accumulate, relu; and the two-stage weight gradient with and without bias."""
    from dgppo_amd import ops_nn as K_
    rng = np.random.default_rng(2024)
    g = torch.Generator().manual_seed(7)
    Ks = [1, 3, 4, 7, 8, 15, 16, 17, 32, 33, 48, 63, 64, 65, 96, 100, 128, 144, 160, 192, 255, 256]
    for trial in range(36):
        Kd = int(rng.choice(Ks)); Nd = int(rng.integers(1, 193)); M = int(rng.choice([1, 5, 31, 32, 33, 64, 257, 1000, 4099]))
        trans, acc, act = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), int(rng.integers(0, 2))
        strided = bool(rng.integers(0, 2)) and Kd % 4 == 0
        Xf = torch.randn(M, Kd + (8 if strided else 0), generator=g)
        X = Xf[:, 4:4 + Kd] if strided else Xf
        W = torch.randn(Nd, Kd, generator=g) if trans else torch.randn(Kd, Nd, generator=g)
        b = torch.randn(Nd, generator=g) if rng.integers(0, 2) else None
        Y0 = torch.randn(M, Nd, generator=g)
        want = X @ (W.T if trans else W) + (b if b is not None else 0.0)
        if acc:
            want = want + Y0
        if act:
            want = torch.relu(want)
        Xd = Xf.to(cuda)[:, 4:4 + Kd] if strided else X.to(cuda)
        Y = Y0.to(cuda).clone() if acc else torch.full((M, Nd), float("nan"), device=cuda)
        K_.dense_fwd(Xd, W.to(cuda), None if b is None else b.to(cuda), Y, act=act, accumulate=acc, trans_w=trans)
        _close(Y, want, 3e-6 * math.sqrt(Kd) + 1e-6, f"dense_fwd M={M} K={Kd} N={Nd} trans={trans} acc={acc} act={act} strided={strided}")
    for trial in range(20):
        Kd = int(rng.choice(Ks)); Nd = int(rng.integers(1, 193)); M = int(rng.choice([1, 31, 64, 65, 300, 2049, 9000]))
        X = torch.randn(M, Kd, generator=g); dY = torch.randn(M, Nd, generator=g)
        dW0 = torch.randn(Kd, Nd, generator=g); db0 = torch.randn(Nd, generator=g)
        use_b = bool(rng.integers(0, 2))
        dW = dW0.to(cuda).clone(); db = db0.to(cuda).clone() if use_b else None
        K_.dense_bwd_w(X.to(cuda), dY.to(cuda), dW, db)
        _close(dW, dW0 + X.T @ dY, 3e-6 * math.sqrt(M) + 1e-6, f"dense_bwd_w M={M} K={Kd} N={Nd}")
        if use_b:
            _close(db, db0 + dY.sum(0), 3e-6 * math.sqrt(M) + 1e-6, f"dense_bwd_w bias M={M} N={Nd}")


@pytest.mark.parametrize("M,save,strided", [(1, True, False), (33, False, False), (1000, True, True), (40000, True, False)])
def test_mlp_gi_fused(cuda, M, save, strided):
    """fused Dense->LN->ReLU->Dense->LN->ReLU->Dense(192) against the plain torch fp32 composition (mlp.py:17-29,
    rnn.py:14-30); ragged M (not a multiple of the 32-row tile), strided input, with and without the saved activations."""
    from dgppo_amd import ops_nn as K_
    g = torch.Generator().manual_seed(M)
    Xfull = torch.randn(M, 80, generator=g)
    X = Xfull[:, 8:72] if strided else Xfull[:, :64].contiguous()
    P = {k: torch.randn(*shp, generator=g) * sc for k, shp, sc in (
        ("W1", (64, 64), 0.2), ("b1", (64,), 0.1), ("g1", (64,), 1.0), ("be1", (64,), 0.1), ("W2", (64, 64), 0.2),
        ("b2", (64,), 0.1), ("g2", (64,), 1.0), ("be2", (64,), 0.1), ("Wi", (64, 192), 0.2), ("bi", (192,), 0.1))}

    def ln(v, gam, bet):   # flax LayerNorm: fast variance, eps 1e-6
        mean = v.mean(-1, keepdim=True)
        var = ((v * v).mean(-1, keepdim=True) - mean * mean).clamp_min(0.0)
        rstd = torch.rsqrt(var + 1e-6)
        return (v - mean) * rstd * gam + bet, mean, rstd

    p1 = X @ P["W1"] + P["b1"]
    o1, m1, r1 = ln(p1, P["g1"], P["be1"])
    y1 = torch.relu(o1)
    p2 = y1 @ P["W2"] + P["b2"]
    o2, m2, r2 = ln(p2, P["g2"], P["be2"])
    y2 = torch.relu(o2)
    gi_want = y2 @ P["Wi"] + P["bi"]
    d = {k: v.to(cuda) for k, v in P.items()}
    Xd = Xfull.to(cuda)[:, 8:72] if strided else X.to(cuda)
    gi = torch.full((M, 192), float("nan"), device=cuda)
    saves = None
    if save:
        saves = tuple(torch.full((M, w), float("nan"), device=cuda) for w in (64, 64, 2, 64, 64, 2))
    K_.mlp_gi_fwd(Xd, d["W1"], d["b1"], d["g1"], d["be1"], d["W2"], d["b2"], d["g2"], d["be2"], d["Wi"], d["bi"], gi, saves)
    _close(gi, gi_want, 2e-5, "gi")
    if save:
        for got, want, nm in zip(saves, (p1, y1, torch.cat([m1, r1], 1), p2, y2, torch.cat([m2, r2], 1)),
                                 ("p1", "y1", "st1", "p2", "y2", "st2")):
            _close(got, want, 2e-5, nm)


@pytest.mark.parametrize("M,masked", [(1, False), (45, True), (1000, False), (33000, True)])
def test_mlp_gi_bwd_fused(cuda, M, masked):
    """dgppo_mlp_gi_bwd (dense^T -> LN+ReLU backward -> dense^T -> LN+ReLU backward -> dense^T in one launch) against autograd of
    the plain torch fp32 composition (jax.grad through mlp.py:17-29 / rnn.py:14-30), and against the five unfused launches it
    replaces; ragged M, with and without the ReLU mask on dx."""
    from dgppo_amd import ops_nn as K_
    g = torch.Generator().manual_seed(M + 3)
    X = torch.randn(M, 64, generator=g).requires_grad_()
    P = {k: (torch.randn(*shp, generator=g) * sc) for k, shp, sc in (
        ("W1", (64, 64), 0.2), ("b1", (64,), 0.1), ("g1", (64,), 1.0), ("be1", (64,), 0.1), ("W2", (64, 64), 0.2),
        ("b2", (64,), 0.1), ("g2", (64,), 1.0), ("be2", (64,), 0.1), ("Wi", (64, 192), 0.2), ("bi", (192,), 0.1))}
    for k in ("g1", "be1", "g2", "be2"):
        P[k].requires_grad_()

    def ln(v, gam, bet):
        mean = v.mean(-1, keepdim=True)
        var = ((v * v).mean(-1, keepdim=True) - mean * mean).clamp_min(0.0)
        rstd = torch.rsqrt(var + 1e-6)
        return (v - mean) * rstd * gam + bet, mean, rstd
    p1 = X @ P["W1"] + P["b1"]; p1.retain_grad()
    o1, m1, r1 = ln(p1, P["g1"], P["be1"]); y1 = torch.relu(o1)
    p2 = y1 @ P["W2"] + P["b2"]; p2.retain_grad()
    o2, m2, r2 = ln(p2, P["g2"], P["be2"]); y2 = torch.relu(o2)
    gi = y2 @ P["Wi"] + P["bi"]
    dgi = torch.randn(M, 192, generator=g)
    gi.backward(dgi)
    mask = torch.randn(M, 64, generator=g) if masked else None
    want_dx = X.grad * (mask > 0) if masked else X.grad
    d = lambda t: t.detach().to(cuda).contiguous()
    outs = {k: torch.full((M, 64), float("nan"), device=cuda) for k in ("dpre2", "dpre1", "dx")}
    pg = {k: torch.zeros(64, device=cuda) for k in ("dg2", "db2", "dg1", "db1")}
    K_.mlp_gi_bwd(d(dgi), d(P["Wi"]), d(P["W2"]), d(P["W1"]), d(P["g2"]), d(P["g1"]), d(p2), d(y2), d(torch.cat([m2, r2], 1)),
                  d(p1), d(y1), d(torch.cat([m1, r1], 1)), d(mask) if masked else None, outs["dpre2"], outs["dpre1"], outs["dx"],
                  pg["dg2"], pg["db2"], pg["dg1"], pg["db1"])
    _close(outs["dpre2"], p2.grad, 2e-5, "dpre2")
    _close(outs["dpre1"], p1.grad, 2e-5, "dpre1")
    _close(outs["dx"], want_dx, 2e-5, "dx")
    for k, want in (("dg2", P["g2"].grad), ("db2", P["be2"].grad), ("dg1", P["g1"].grad), ("db1", P["be1"].grad)):
        _close(pg[k], want, 3e-5, k)
    # the unfused sequence of launches gives the same numbers
    dy = torch.empty(M, 64, device=cuda); dp2 = torch.empty(M, 64, device=cuda); dp1 = torch.empty(M, 64, device=cuda)
    dxu = torch.empty(M, 64, device=cuda)
    ug = {k: torch.zeros(64, device=cuda) for k in ("dg2", "db2", "dg1", "db1")}
    K_.dense_fwd(d(dgi), d(P["Wi"]), None, dy, trans_w=True)
    K_.ln_relu_bwd(d(p2), d(y2), d(torch.cat([m2, r2], 1)), d(P["g2"]), dy, dp2, ug["dg2"], ug["db2"])
    K_.dense_fwd(dp2, d(P["W2"]), None, dy, trans_w=True)
    K_.ln_relu_bwd(d(p1), d(y1), d(torch.cat([m1, r1], 1)), d(P["g1"]), dy, dp1, ug["dg1"], ug["db1"])
    K_.dense_fwd(dp1, d(P["W1"]), None, dxu, trans_w=True, relu_mask=d(mask) if masked else None)
    _close(outs["dx"], dxu.cpu(), 2e-5, "dx vs unfused")
    _close(outs["dpre2"], dp2.cpu(), 2e-5, "dpre2 vs unfused")


@pytest.mark.parametrize("M,two,use_h0,save", [(1, True, True, True), (45, False, False, False), (1000, True, False, True),
                                               (33000, False, True, True), (33000, True, True, False)])
def test_gru1_head_fused(cuda, M, two, use_h0, save):
    """fused one-step GRU + head Dense(s) against the plain torch fp32 composition (flax GRUCell gate order r|z|n,
    rnn.py:14-30; policy.py:62-74 / value.py:41,76), ragged M, zero or given initial carry, with and without saves."""
    from dgppo_amd import ops_nn as K_
    g = torch.Generator().manual_seed(M + 7)
    n_out = 4 if two else 2
    gi = torch.randn(M, 192, generator=g)
    h0 = torch.randn(M, 64, generator=g) if use_h0 else None
    Wh = torch.randn(64, 192, generator=g) * 0.2
    bhn = torch.randn(64, generator=g) * 0.1
    if two:
        W1, b1 = torch.randn(64, 64, generator=g) * 0.2, torch.randn(64, generator=g) * 0.1
        W2, b2 = torch.randn(64, n_out, generator=g) * 0.2, torch.randn(n_out, generator=g) * 0.1
    else:
        W1, b1, W2, b2 = torch.randn(64, n_out, generator=g) * 0.2, torch.randn(n_out, generator=g) * 0.1, None, None
    h = h0 if use_h0 else torch.zeros(M, 64)
    gh = h @ Wh
    r = torch.sigmoid(gi[:, :64] + gh[:, :64])
    z = torch.sigmoid(gi[:, 64:128] + gh[:, 64:128])
    hn = gh[:, 128:] + bhn
    nn_ = torch.tanh(gi[:, 128:] + r * hn)
    hnew = (1 - z) * nn_ + z * h
    if two:
        u_want = hnew @ W1 + b1
        out_want = u_want @ W2 + b2
    else:
        u_want, out_want = None, hnew @ W1 + b1
    d = lambda t: None if t is None else t.to(cuda)
    hs = torch.full((M, 64), float("nan"), device=cuda)
    out = torch.full((M, n_out), float("nan"), device=cuda)
    hprev = torch.full((M, 64), float("nan"), device=cuda) if save else None
    gates = torch.full((M, 256), float("nan"), device=cuda) if save else None
    u = torch.full((M, 64), float("nan"), device=cuda) if (save and two) else None
    K_.gru1_head_fwd(d(gi), d(Wh), d(bhn), d(h0), d(W1), d(b1), d(W2), d(b2), hs, hprev, gates, u, out)
    _close(hs, hnew, 1e-5, "hs")
    _close(out, out_want, 2e-5, "out")
    if save:
        _close(hprev, h, 1e-6, "hprev")
        _close(gates, torch.cat([r, z, nn_, hn], 1), 1e-5, "gates")
        if two:
            _close(u, u_want, 2e-5, "u")


def test_ln_relu(cuda):
    from dgppo_amd import ops_nn as K_
    g = torch.Generator().manual_seed(0)
    M = 1037
    x = (torch.randn(M, 64, generator=g) * 2 + 0.3).requires_grad_()
    p = {"scale": (torch.rand(64, generator=g) + 0.5).requires_grad_(), "bias": (torch.randn(64, generator=g) * 0.1).requires_grad_()}
    y = torch.relu(T.layer_norm(p, x))
    dy = torch.randn(M, 64, generator=g)
    y.backward(dy)
    xd = x.detach().to(cuda)
    yd = torch.empty(M, 64, device=cuda)
    st = torch.empty(M, 2, device=cuda)
    K_.ln_relu_fwd(xd, p["scale"].detach().to(cuda), p["bias"].detach().to(cuda), yd, st)
    _close(yd, y, 1e-5, "ln fwd")
    dx = torch.empty(M, 64, device=cuda)
    dg = torch.zeros(64, device=cuda)
    db = torch.zeros(64, device=cuda)
    K_.ln_relu_bwd(xd, yd, st, p["scale"].detach().to(cuda), dy.to(cuda), dx, dg, db)
    _close(dx, x.grad, 2e-5, "ln dx")
    _close(dg, p["scale"].grad, 2e-5, "ln dgamma")
    _close(db, p["bias"].grad, 2e-5, "ln dbeta")


@pytest.mark.parametrize("n_grp,T_,n_inner,use_h0", [(5, 16, 8, False), (70, 1, 3, True), (9, 7, 1, True), (130, 3, 1, False),
                                                     (1, 129, 1, False),       # one sequence, the pre-pass length
                                                     (2051, 2, 8, True),       # 16 408 sequences: the 32-sequence-tile kernels
                                                     (16390, 1, 1, True)])     # same kernels, T = 1, ragged last tile
def test_gru_scan(cuda, n_grp, T_, n_inner, use_h0):
    from dgppo_amd import ops_nn as K_
    g = torch.Generator().manual_seed(n_grp + T_)
    p = T.init_gru(g)
    p["hn"]["bias"] = torch.randn(64, generator=g) * 0.1
    for k in p:
        for kk in p[k]:
            p[k][kk].requires_grad_()
    n_seq = n_grp * n_inner
    rows = n_seq * T_
    x = (torch.randn(rows, 64, generator=g)).requires_grad_()      # row = (grp*T + tau)*n_inner + i
    h0 = (torch.randn(n_seq, 64, generator=g) * 0.5).requires_grad_() if use_h0 else None
    xs = x.view(n_grp, T_, n_inner, 64)
    h = h0.view(n_grp, n_inner, 64) if use_h0 else torch.zeros(n_grp, n_inner, 64)
    outs = []
    for tau in range(T_):
        h = T.gru_cell(p, h, xs[:, tau])
        outs.append(h)
    hs = torch.stack(outs, 1).reshape(rows, 64)
    dhs = torch.randn(rows, 64, generator=g)
    hs.backward(dhs)
    Wi = torch.cat([p[k]["kernel"] for k in ("ir", "iz", "in")], 1).detach()
    bi = torch.cat([p[k]["bias"] for k in ("ir", "iz", "in")]).detach()
    Wh = torch.cat([p[k]["kernel"] for k in ("hr", "hz", "hn")], 1).detach().contiguous().to(cuda)
    bhn = p["hn"]["bias"].detach().to(cuda)
    gi = (x.detach() @ Wi + bi).to(cuda)
    hs_d = torch.empty(rows, 64, device=cuda)
    hprev = torch.empty(rows, 64, device=cuda)
    gates = torch.empty(rows, 256, device=cuda)
    K_.gru_fwd(gi, Wh, bhn, h0.detach().to(cuda) if use_h0 else None, hs_d, hprev, gates, n_seq, T_, n_inner)
    _close(hs_d, hs, 1e-5, "gru hs")
    dgi = torch.empty(rows, 192, device=cuda)
    dgh = torch.empty(rows, 192, device=cuda)
    K_.gru_bwd(dhs.to(cuda), Wh, hprev, gates, dgi, dgh, n_seq, T_, n_inner)
    # dx = dgi Wi^T ; dWi = x^T dgi ; dWh = hprev^T dgh ; dbhn = colsum dgh[:,128:]
    _close(dgi.cpu() @ Wi.T, x.grad, 2e-5, "gru dx")
    dWi = x.detach().T @ dgi.cpu()
    _close(dWi[:, :64], p["ir"]["kernel"].grad, 3e-5, "dWir")
    _close(dWi[:, 128:], p["in"]["kernel"].grad, 3e-5, "dWin")
    dWh = hprev.cpu().T @ dgh.cpu()
    _close(dWh[:, 64:128], p["hz"]["kernel"].grad, 3e-5, "dWhz")
    _close(dWh[:, 128:], p["hn"]["kernel"].grad, 3e-5, "dWhn")
    _close(dgh.cpu()[:, 128:].sum(0), p["hn"]["bias"].grad, 3e-5, "dbhn")
    _close(dgi.cpu().sum(0)[:64], p["ir"]["bias"].grad, 3e-5, "dbir")


def test_policy_head_all_modes(cuda):
    from dgppo_amd import ops_nn as K_
    g = torch.Generator().manual_seed(3)
    n = 8
    rows = 64 * n + 3 * n
    ms = torch.randn(rows, 4, generator=g)
    ms[:, :2] *= 1.5
    ms[5, 0] = 6.0      # drives an action past the 0.999 clip -> log-cdf branch
    ms[6, 1] = -6.0
    eps = torch.randn(rows, 2, generator=g)
    mean, std = ms[:, :2], torch.nn.functional.softplus(ms[:, 2:] + T.STD_INIT_INV) + T.STD_MIN
    act_want = torch.tanh(mean + std * eps)
    lp_want = T.tanh_normal_log_prob(act_want, mean, std)
    msd = ms.to(cuda)
    act = torch.empty(rows, 2, device=cuda)
    lp = torch.empty(rows, device=cuda)
    K_.policy_head(msd, eps.to(cuda), None, act, lp, None, n, 0)
    _close(act, act_want, 1e-6, "sample action")
    assert (act_want.abs() >= 0.999).any()
    _close(lp, lp_want, 1e-5, "sample log_pi")
    K_.policy_head(msd, None, None, act, None, None, n, 1)
    _close(act, torch.tanh(mean), 1e-6, "mode action")
    # eval + PPO loss gradient
    eps_hat = torch.randn(n, 2, generator=g)
    a_in = act_want.clone()
    a_in[10:20] = torch.tanh(torch.randn(10, 2, generator=g))
    lp_old = lp_want + 0.3 * torch.randn(rows, generator=g)
    adv = torch.randn(rows, generator=g)
    msr = ms.clone().requires_grad_()
    mean_r, std_r = msr[:, :2], torch.nn.functional.softplus(msr[:, 2:] + T.STD_INIT_INV) + T.STD_MIN
    lp_r = T.tanh_normal_log_prob(a_in, mean_r, std_r)
    ent_r = T.tanh_normal_entropy(mean_r, std_r, eps_hat.repeat(rows // n, 1))
    rho = torch.exp(lp_r - lp_old)
    l1, l2 = -rho * adv, -torch.clamp(rho, 0.75, 1.25) * adv
    loss = torch.maximum(l1, l2).mean() - 0.01 * ent_r.mean()
    loss.backward()
    lp2 = torch.empty(rows, device=cuda)
    ent2 = torch.empty(rows, device=cuda)
    dms = torch.empty(rows, 4, device=cuda)
    stats = torch.zeros(8, device=cuda)
    K_.policy_head(msd, eps_hat.to(cuda), a_in.to(cuda), None, lp2, ent2, n, 2, lp_old.to(cuda), adv.to(cuda), dms, stats, 0.25, 0.01)
    _close(lp2, lp_r, 1e-5, "eval log_pi")
    _close(ent2, ent_r, 1e-5, "eval entropy")
    _close(dms, msr.grad, 2e-5, "dms")
    s = stats.cpu()
    _close(s[0] / rows - 0.01 * s[1] / rows, loss, 1e-5, "loss")
    _close(s[2] / rows, (l2 > l1).float().mean(), 1e-6, "clip_frac")
    _close(0.5 * s[3] / rows, 0.5 * (rho - 1).abs().mean(), 1e-5, "tv")


# ----------------------------------------------------------------------------------------------------------------------
# composed networks on real graphs
# ----------------------------------------------------------------------------------------------------------------------
def _scene(kind, n, n_obs, n_env, T_steps, seed):
    """roll the oracle env for T_steps with random actions; returns per-(env,t) compact records + materialised graphs."""
    from dgppo_amd import _native as N
    ocfg = E.EnvCfg(kind, n_agents=n, n_obs=n_obs)
    cfg = N.make_env_cfg(kind, n, n_obs)
    rng = np.random.default_rng(seed)
    agent, goal, obst = E.env_reset(ocfg, rng.integers(1, 2 ** 60, size=n_env))
    agent[:, :, :2] = (agent[:, :, :2] * 0.5 + 0.4).astype(np.float32)     # crowd them so masks vary
    tab = E.ray_table(32)
    hits = None
    if ocfg.is_lidar and n_obs > 0:
        hits, _ = E.lidar_sense(ocfg, agent[..., :2], obst, *tab)
    agents, hitss, graphs = [agent], [hits], [E.get_graph(ocfg, agent, goal, obst, hits)]
    for t in range(T_steps - 1):
        a = rng.uniform(-1, 1, size=(n_env, n, 2)).astype(np.float32)
        out = E.env_step(ocfg, agent, goal, obst, hits, a, tab)
        agent, hits = out["next_agent"], out["next_hits"]
        agents.append(agent); hitss.append(hits); graphs.append(out["graph"])
    ag = np.stack(agents, 1)                                             # [n_env, T, n, sd]
    hi = np.stack(hitss, 1) if hits is not None else None                # [n_env, T, n, k, 2]
    gr = {k: np.stack([g[k] for g in graphs], 1).reshape((n_env * T_steps,) + graphs[0][k].shape[1:]) for k in graphs[0]}
    return cfg, ocfg, ag, goal, obst, hi, gr


def _feats(cfg, ag, goal, obst, hi, dev, tag="t"):
    from dgppo_amd import nets
    n_env, T_steps = ag.shape[:2]
    arena = nets.Arena(dev)
    f = nets.GraphFeats(cfg, n_env * T_steps, arena, tag)
    agd = torch.from_numpy(ag).to(dev)
    hid = torch.from_numpy(hi).to(dev) if hi is not None else None
    n, sd = cfg.n_agents, cfg.state_dim
    f.compute(agd, T_steps * n * sd, n * sd, torch.from_numpy(goal).to(dev),
              torch.from_numpy(obst).to(dev) if obst is not None else None,
              hid, T_steps * n * cfg.top_k * 2, n * cfg.top_k * 2, None, n_env, T_steps)
    f._keep = (agd, hid, arena)
    return f


def _grad_tree_close(net, tree, tol, skip_bk=True):
    gt = net.to_tree(net.grads)
    want = dict(T.tree_leaves(T.tree_map(lambda t: t.grad if t.grad is not None else torch.zeros_like(t), tree)))
    got = dict(T.tree_leaves(T.tree_map(lambda a: torch.from_numpy(np.ascontiguousarray(a)), gt)))
    assert set(got) == set(want)
    gscale = max(float(v.abs().max()) for v in want.values())
    report, bad = [], []
    for k in sorted(want):
        err = float((got[k].double() - want[k].double()).abs().max())
        report.append(f"{k:70s} want_max {float(want[k].abs().max()):.3e} got_max {float(got[k].abs().max()):.3e} err {err:.3e}")
        if skip_bk and "GraphTransformer_" in k and k.endswith("Dense_1/bias"):
            # key bias cancels in the softmax: exact 0 here, fp noise in an autograd of the per-edge form
            if float(got[k].abs().max()) != 0.0 or float(want[k].abs().max()) > 1e-4 * max(gscale, 1.0):
                bad.append(k)
            continue
        if err > tol * max(gscale, 1e-3):
            bad.append(k)
    if bad:
        print("\n".join(report))
    assert not bad, f"gradient mismatch in {bad} (scale {gscale:.3e})"


def _leafify(tree):
    return T.tree_map(lambda t: t.clone().requires_grad_(), tree)


@pytest.mark.parametrize("kind,n,n_obs", [(E.LIDAR_SPREAD, 8, 3), (E.MPE_TARGET, 3, 3), (E.LIDAR_BICYCLE_TARGET, 4, 2),
                                           (E.MPE_SPREAD, 3, 0), (E.LIDAR_BICYCLE_TARGET, 16, 8)])   # last: config 5 size
def test_policy_forward_backward(cuda, kind, n, n_obs):
    from dgppo_amd import nets, ops_nn as K_
    n_env, T_ = 4, 16
    cfg, ocfg, ag, goal, obst, hi, gr = _scene(kind, n, n_obs, n_env, T_, seed=kind * 10 + n)
    tree = T.init_policy(1, cfg.node_dim)
    # non-trivial biases / LayerNorm params so every gradient path is exercised
    gen = torch.Generator().manual_seed(5)
    tree = T.tree_map(lambda t: t + 0.05 * torch.randn(t.shape, generator=gen), tree)
    # ScaleHid is 0.01 * orthogonal at init: rescale so mean / std_trans are O(0.3) and every branch is exercised
    tree["params"]["ScaleHid"]["kernel"] = T.orthogonal(gen, 64, 64, 0.5)
    net = nets.Net("policy", cfg, 2, 2, cuda)
    net.load_tree(tree)
    feats = _feats(cfg, ag, goal, obst, hi, cuda)
    G = n_env * T_
    act = net.forward(feats, n_seq=n_env * n, T=T_, h0=None)
    # oracle: scan over the chunk with zero initial carry (informarl.py:409-424)
    lt = _leafify(tree)
    g_t = T.graph_to_torch(gr)
    gsel = lambda t: {k: v.view((n_env, T_) + v.shape[1:])[:, t] for k, v in g_t.items()}
    rng = torch.Generator().manual_seed(9)
    a_in = torch.tanh(torch.randn(n_env, T_, n, 2, generator=rng))
    eps_hat = torch.randn(n, 2, generator=rng)
    h = torch.zeros(n_env, n, 64)
    lps, ents, hss = [], [], []
    for t in range(T_):
        lp, ent, h = T.policy_eval(lt, gsel(t), a_in[:, t], h, n, eps_hat)
        lps.append(lp); ents.append(ent); hss.append(h)
    lp_w, ent_w, hs_w = torch.stack(lps, 1), torch.stack(ents, 1), torch.stack(hss, 1)
    _close(act["hs"].view(n_env, T_, n, 64), hs_w, 1e-5, "policy hidden")
    R = G * n
    lp_old = (lp_w.detach() + 0.2 * torch.randn(n_env, T_, n, generator=rng))
    adv = torch.randn(n_env, T_, n, generator=rng)
    rho = torch.exp(lp_w - lp_old)
    loss = torch.maximum(-rho * adv, -torch.clamp(rho, 0.75, 1.25) * adv).mean() - 0.01 * ent_w.mean()
    loss.backward()
    lp = torch.empty(R, device=cuda); ent = torch.empty(R, device=cuda)
    dms = torch.empty(R, 4, device=cuda); stats = torch.zeros(8, device=cuda)
    K_.policy_head(act["ms"], eps_hat.to(cuda), a_in.reshape(R, 2).to(cuda), None, lp, ent, n, 2,
                   lp_old.reshape(R).to(cuda), adv.reshape(R).to(cuda), dms, stats, 0.25, 0.01)
    _close(lp.view(n_env, T_, n), lp_w, 1e-5, "policy log_pi")
    _close(ent.view(n_env, T_, n), ent_w, 1e-5, "policy entropy")
    net.zero_grads()
    net.backward(act, dms)
    torch.cuda.synchronize()
    _grad_tree_close(net, lt, 3e-5)


@pytest.mark.parametrize("kind,n,n_obs", [(E.LIDAR_SPREAD, 8, 3), (E.MPE_SPREAD, 3, 3)])
def test_Vl_forward_backward(cuda, kind, n, n_obs):
    from dgppo_amd import nets, ops_nn as K_
    n_env, T_ = 5, 8
    cfg, ocfg, ag, goal, obst, hi, gr = _scene(kind, n, n_obs, n_env, T_, seed=3)
    gen = torch.Generator().manual_seed(6)
    tree = T.tree_map(lambda t: t + 0.05 * torch.randn(t.shape, generator=gen), T.init_value(2, cfg.node_dim, 1, 2))
    net = nets.Net("Vl", cfg, 2, 1, cuda)
    net.load_tree(tree)
    feats = _feats(cfg, ag, goal, obst, hi, cuda)
    act = net.forward(feats, n_seq=n_env, T=T_, h0=None)
    lt = _leafify(tree)
    g_t = T.graph_to_torch(gr)
    gsel = lambda t: {k: v.view((n_env, T_) + v.shape[1:])[:, t] for k, v in g_t.items()}
    h = torch.zeros(n_env, 1, 64)
    vs = []
    for t in range(T_):
        v, h = T.value_Vl(lt, gsel(t), h, n)
        vs.append(v)
    v_w = torch.stack(vs, 1)
    _close(act["v"].view(n_env, T_), v_w, 1e-5, "Vl")
    target = torch.randn(n_env, T_, generator=gen)
    (0.5 * (v_w - target) ** 2).mean().backward()
    dv = torch.empty(n_env * T_, 1, device=cuda)
    stats = torch.zeros(8, device=cuda)
    K_.value_loss(act["v"], target.reshape(-1, 1).to(cuda), dv, stats)
    net.zero_grads()
    net.backward(act, dv)
    torch.cuda.synchronize()
    _close(stats[0].cpu() / (n_env * T_), (0.5 * (v_w - target) ** 2).mean(), 1e-5, "Vl loss")
    _grad_tree_close(net, lt, 3e-5)


@pytest.mark.parametrize("kind,n,n_obs", [(E.LIDAR_SPREAD, 8, 3), (E.LIDAR_TARGET, 3, 2)])
def test_Vh_forward_backward(cuda, kind, n, n_obs):
    from dgppo_amd import nets, ops_nn as K_
    n_env, T_ = 3, 6
    cfg, ocfg, ag, goal, obst, hi, gr = _scene(kind, n, n_obs, n_env, T_, seed=4)
    gen = torch.Generator().manual_seed(7)
    tree = T.tree_map(lambda t: t + 0.05 * torch.randn(t.shape, generator=gen), T.init_value(3, cfg.node_dim, 2, 1))
    net = nets.Net("Vh", cfg, 1, 2, cuda)
    net.load_tree(tree)
    feats = _feats(cfg, ag, goal, obst, hi, cuda)
    G = n_env * T_
    h0 = torch.randn(G, n, 64, generator=gen) * 0.5          # the actor's stored carry (dgppo.py:128-134,219-220)
    act = net.forward(feats, n_seq=G * n, T=1, h0=h0.reshape(G * n, 64).to(cuda))
    lt = _leafify(tree)
    v_w, _ = T.value_Vh(lt, T.graph_to_torch(gr), h0, n)
    _close(act["v"].view(G, n, 2), v_w, 1e-5, "Vh")
    target = torch.randn(G, n, 2, generator=gen)
    (0.5 * (v_w - target) ** 2).mean().backward()
    dv = torch.empty(G * n, 2, device=cuda)
    stats = torch.zeros(8, device=cuda)
    K_.value_loss(act["v"], target.reshape(-1, 2).to(cuda), dv, stats)
    net.zero_grads()
    net.backward(act, dv)
    torch.cuda.synchronize()
    _grad_tree_close(net, lt, 3e-5)


def test_tree_roundtrip_and_param_counts(cuda):
    """parameter counts of SURVEY A.9: 62 660 / 58 305 / 39 426 floats at node_dim 7."""
    from dgppo_amd import nets, _native as N
    cfg = N.make_env_cfg(0, 8, 3)
    for kind, layers, n_out, count, tree in (("policy", 2, 2, 62660, T.init_policy(0, 7)),
                                             ("Vl", 2, 1, 58305, T.init_value(0, 7, 1, 2)),
                                             ("Vh", 1, 2, 39426, T.init_value(0, 7, 2, 1))):
        net = nets.Net(kind, cfg, layers, n_out, cuda)
        net.load_tree(tree)
        back = net.to_tree()
        leaves_a = dict(T.tree_leaves(tree))
        leaves_b = dict(T.tree_leaves(back))
        assert set(leaves_a) == set(leaves_b)
        assert sum(v.numel() for v in leaves_a.values()) == count
        for k in leaves_a:
            np.testing.assert_array_equal(leaves_a[k].numpy(), leaves_b[k])


@pytest.mark.parametrize("F,Kp", [(8, 48), (32, 144), (16, 80)])
def test_attention_kernel_families_agree(cuda, monkeypatch, F, Kp):
    """dgppo_attn_fwd/bwd dispatch between the one-wave-per-graph kernels (default for F in {8, 32}), the workgroup-per-
    graph MFMA kernels and the VALU kernels.  The default path is pinned to the oracle by the
    network tests below; this pins every fallback to the default path on the same random inputs (masked slots, NaN edge
    features behind the mask).  F = 16 has no wave instantiation: there the MFMA and VALU families are compared."""
    from dgppo_amd import _native as N, ops_nn as K_
    cfg = N.make_env_cfg(0, 8, 3)
    n, S, H = 8, cfg.fan_in, 3
    n_other = cfg.num_nodes - 1 - n
    G = 37
    g = torch.Generator().manual_seed(F)
    R = G * n
    qt = torch.randn(R, H * F, generator=g).to(cuda)
    Xa = torch.randn(R, F, generator=g).to(cuda)
    Xo = torch.randn(G * n_other, F, generator=g).to(cuda)
    em = (torch.rand(R, S, generator=g) > 0.35).float()
    em[:, :n] = 1.0
    ef = torch.randn(R, S, 4, generator=g)
    ef[em == 0] = float("nan")                 # masked slots must never be multiplied
    em, ef = em.to(cuda), ef.to(cuda)
    dz = torch.randn(R, Kp, generator=g).to(cuda)

    def run():
        z = torch.full((R, Kp), float("nan"), device=cuda)
        at = torch.full((R, S, H), float("nan"), device=cuda)
        K_.attn_fwd(cfg, F, H, Kp, qt, Xa, Xo, ef, em, z, at, G)
        dq = torch.full((R, H * F), float("nan"), device=cuda)
        dXa = torch.full((R, F), float("nan"), device=cuda)
        dXo = torch.full((G * n_other, F), float("nan"), device=cuda)
        K_.attn_bwd(cfg, F, H, Kp, dz, at, qt, Xa, Xo, ef, dq, dXa, dXo, G)
        # relu_xo: the same call with the ReLU backward of the other nodes' gradient fused in (dXo *= (Xo > 0))
        dq2, dXa2 = torch.empty_like(dq), torch.empty_like(dXa)
        dXo2 = torch.full_like(dXo, float("nan"))
        K_.attn_bwd(cfg, F, H, Kp, dz, at, qt, Xa, Xo, ef, dq2, dXa2, dXo2, G, relu_xo=True)
        # first-layer form: no input gradient requested (F = 8 takes the slot-sparse VALU kernel by default)
        dq3 = torch.full((R, H * F), float("nan"), device=cuda)
        K_.attn_bwd(cfg, F, H, Kp, dz, at, qt, Xa, Xo, ef, dq3, None, None, G)
        torch.cuda.synchronize()
        assert torch.equal(dq2, dq) and torch.equal(dXa2, dXa)
        assert torch.equal(dXo2, torch.where(Xo > 0, dXo, torch.zeros_like(dXo))), "relu_xo must equal masking afterwards"
        return dict(z=z, at=at, dq=dq, dXa=dXa, dXo=dXo, dq_only=dq3)

    families = {"default": {}, "block": {"DGPPO_ATTN_BLOCK": "1"}, "valu": {"DGPPO_ATTN_VALU": "1"},
                "dense8": {"DGPPO_ATTN_DENSE8": "1"},      # F = 8: the matrix-core wave kernels instead of the slot-sparse ones
                "wave": {"DGPPO_ATTN_NO_BD": "1"},         # F = 32: the dense one-wave-per-graph tiles instead of the block-diagonal form
                "persist": {"DGPPO_ATTN_PERSIST_WGS": "3"}}  # F = 32 forward: persistent waves, 6 graphs each, next graph's loads in flight
    outs = {}
    for name, env in families.items():
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        outs[name] = run()
        for k in env:
            monkeypatch.delenv(k)
    ref = outs["default"]
    for k, v in ref.items():
        assert torch.isfinite(v).all(), f"default path left non-finite values in {k}"
    _close(ref["dq_only"], ref["dq"], 2e-5, "dqt-only backward vs full backward")
    for name in ("block", "valu", "dense8", "wave", "persist"):
        for k in ref:
            _close(outs[name][k], ref[k], 2e-5, f"{name}.{k}")


@pytest.mark.parametrize("kind,n,n_obs", [(0, 2, 1), (0, 3, 0), (0, 5, 2), (0, 10, 3), (1, 6, 2), (3, 4, 3), (4, 6, 5), (2, 4, 2),
                                           (0, 8, 5), (1, 10, 1)])
def test_attention_wave_kernels_across_topologies(cuda, monkeypatch, kind, n, n_obs):
    """the one-wave-per-graph kernels are compiled per (node tiles CT, softmax passes NP, slots per lane SJ): walk
    different environment topologies (fan-in 2..35, 3..71 nodes) and compare forward and backward with the VALU
    kernels, whose code does not depend on those parameters (shapes without a wave instantiation take a fallback on both
    sides, which then simply must agree with itself)."""
    from dgppo_amd import _native as N, ops_nn as K_
    cfg = N.make_env_cfg(kind, n, n_obs)
    S, H = cfg.fan_in, 3
    n_other = cfg.num_nodes - 1 - n
    G = 21
    gen = torch.Generator().manual_seed(kind * 100 + n * 10 + n_obs)
    R = G * n
    for F, Kp in ((8, 48), (32, 144)):
        qt = torch.randn(R, H * F, generator=gen).to(cuda)
        Xa = torch.randn(R, F, generator=gen).to(cuda)
        Xo = torch.randn(max(G * n_other, 1), F, generator=gen).to(cuda)[:G * n_other]
        em = (torch.rand(R, S, generator=gen) > 0.3).float()
        em[:, 0] = 1.0
        ef = torch.randn(R, S, 4, generator=gen)
        ef[em == 0] = float("nan")
        em, ef = em.to(cuda), ef.to(cuda)
        dz = torch.randn(R, Kp, generator=gen).to(cuda)

        def run():
            z = torch.full((R, Kp), float("nan"), device=cuda)
            at = torch.full((R, S, H), float("nan"), device=cuda)
            K_.attn_fwd(cfg, F, H, Kp, qt, Xa, Xo if n_other > 0 else None, ef, em, z, at, G)
            dq = torch.full((R, H * F), float("nan"), device=cuda)
            dXa = torch.full((R, F), float("nan"), device=cuda)
            dXo = torch.full((G * n_other, F), float("nan"), device=cuda) if n_other > 0 else None
            K_.attn_bwd(cfg, F, H, Kp, dz, at, qt, Xa, Xo if n_other > 0 else None, ef, dq, dXa, dXo, G)
            dq3 = torch.full((R, H * F), float("nan"), device=cuda)           # first-layer form (slot-sparse kernel for F = 8)
            K_.attn_bwd(cfg, F, H, Kp, dz, at, qt, Xa, Xo if n_other > 0 else None, ef, dq3, None, None, G)
            torch.cuda.synchronize()
            return dict(z=z, at=at, dq=dq, dXa=dXa, dq_only=dq3, **({"dXo": dXo} if dXo is not None else {}))

        got = run()
        monkeypatch.setenv("DGPPO_ATTN_VALU", "1")
        ref = run()
        monkeypatch.delenv("DGPPO_ATTN_VALU")
        for k in ref:
            assert torch.isfinite(got[k]).all(), f"{k} not finite (F={F})"
            _close(got[k], ref[k], 2e-5, f"F={F} {k}")


@pytest.mark.parametrize("kind,n,n_obs", [(0, 8, 3), (0, 3, 0), (0, 10, 3), (1, 6, 2), (3, 4, 3), (2, 16, 8)])
def test_attention_with_recomputed_other_nodes(cuda, monkeypatch, kind, n, n_obs):
    """dgppo_attn_fwd_xo / _bwd_xo recompute the sender rows of the nodes without incoming edges, relu(Xo_raw Wo + bo), inside
    the kernel (gnn.py:109-111 with aggr = 0 feeding gnn.py:85-117); they must agree with materialising those rows (torch) and
    calling dgppo_attn_fwd / _bwd, forward and backward, with and without the ReLU mask on dXo."""
    from dgppo_amd import _native as N, ops_nn as K_
    cfg = N.make_env_cfg(kind, n, n_obs)
    F, Kp, H, S = 32, 144, 3, cfg.fan_in
    n_other = cfg.num_nodes - 1 - n
    if not K_.attn_xo_supported(cfg, F, H, Kp):
        pytest.skip("no fused kernel for this topology")
    G = 19
    gen = torch.Generator().manual_seed(kind * 100 + n * 10 + n_obs + 7)
    R = G * n
    qt = torch.randn(R, H * F, generator=gen).to(cuda)
    Xa = torch.randn(R, F, generator=gen).to(cuda)
    raw = torch.randn(G * n_other, 8, generator=gen).to(cuda)
    Wfull = (torch.randn(Kp, 32, generator=gen) * 0.5).to(cuda)         # the rows [:8] of a wider matrix, as in the network
    Wo, bo = Wfull[:8], (torch.randn(32, generator=gen) * 0.3).to(cuda)
    Xo = torch.relu(raw @ Wo + bo).contiguous()
    em = (torch.rand(R, S, generator=gen) > 0.3).float()
    em[:, 0] = 1.0
    ef = torch.randn(R, S, 4, generator=gen)
    ef[em == 0] = float("nan")
    em, ef = em.to(cuda), ef.to(cuda)
    dz = torch.randn(R, Kp, generator=gen).to(cuda)

    def run(fused, relu_xo):
        z = torch.full((R, Kp), float("nan"), device=cuda)
        at = torch.full((R, S, H), float("nan"), device=cuda)
        dq = torch.full((R, H * F), float("nan"), device=cuda)
        dXa = torch.full((R, F), float("nan"), device=cuda)
        dXo = torch.full((G * n_other, F), float("nan"), device=cuda)
        if fused:
            K_.attn_fwd_xo(cfg, F, H, Kp, qt, Xa, raw, Wo, bo, ef, em, z, at, G)
            K_.attn_bwd_xo(cfg, F, H, Kp, dz, at, qt, Xa, raw, Wo, bo, ef, dq, dXa, dXo, G, relu_xo=relu_xo)
            z2 = torch.full((R, Kp), float("nan"), device=cuda)
            K_.attn_fwd_xo(cfg, F, H, Kp, qt, Xa, raw, Wo, bo, ef, em, z2, None, G)       # inference form: no weights kept
            assert torch.equal(z2, z)
        else:
            K_.attn_fwd(cfg, F, H, Kp, qt, Xa, Xo, ef, em, z, at, G)
            K_.attn_bwd(cfg, F, H, Kp, dz, at, qt, Xa, Xo, ef, dq, dXa, dXo, G, relu_xo=relu_xo)
        torch.cuda.synchronize()
        return dict(z=z, at=at, dq=dq, dXa=dXa, dXo=dXo)

    monkeypatch.setenv("DGPPO_ATTN_PERSIST_WGS", "2")            # the persistent forward (4-5 graphs per wave) on the same inputs
    pers = run(True, False)
    monkeypatch.delenv("DGPPO_ATTN_PERSIST_WGS")
    plain = run(True, False)
    for k in plain:
        assert torch.equal(pers[k], plain[k]), f"persistent forward differs in {k}"
    for relu_xo in (False, True):
        got, ref = run(True, relu_xo), run(False, relu_xo)
        for k in ref:
            assert torch.isfinite(got[k]).all(), f"{k} not finite"
            _close(got[k], ref[k], 2e-5, f"relu_xo={relu_xo} {k}")
    # the variant that consumes the gradient of the recomputed rows: dWo += raw^T (relu' dXo), dbo += its column sums
    ref = run(False, True)
    at = torch.empty(R, S, H, device=cuda)
    z = torch.empty(R, Kp, device=cuda)
    K_.attn_fwd_xo(cfg, F, H, Kp, qt, Xa, raw, Wo, bo, ef, em, z, at, G)
    dq = torch.full((R, H * F), float("nan"), device=cuda)
    dXa = torch.full((R, F), float("nan"), device=cuda)
    dWfull = torch.full((Kp, 32), 0.25, device=cuda)              # accumulated into, rows [:8] of a wider gradient
    dbo = torch.full((32,), -0.5, device=cuda)
    ws = torch.empty(K_.attn_xo_workspace_floats(G), device=cuda)
    K_.attn_bwd_xo_dw(cfg, F, H, Kp, dz, at, qt, Xa, raw, Wo, bo, ef, dq, dXa, dWfull[:8], dbo, ws, G)
    torch.cuda.synchronize()
    _close(dq, ref["dq"], 2e-5, "dw variant dq")
    _close(dXa, ref["dXa"], 2e-5, "dw variant dXa")
    dXo64 = ref["dXo"].double()
    _close(dWfull[:8], (0.25 + raw.double().t() @ dXo64).float(), 2e-5, "dWo")
    _close(dbo, (-0.5 + dXo64.sum(0)).float(), 2e-5, "dbo")
    assert torch.equal(dWfull[8:], torch.full((Kp - 8, 32), 0.25, device=cuda)), "rows past 8 of the wider gradient were touched"


@pytest.mark.parametrize("M,K,N,acc", [(1000, 96, 32, True), (513, 64, 64, False), (70, 24, 8, True)])
def test_dense_fwd_relu_mask_epilogue(cuda, M, K, N, acc):
    """dgppo_dense_fwd(relu_mask=...): Y = where(mask > 0, X W^T (+ Y), 0) — the ReLU backward fused into the kernel that
    finishes a gradient — must equal the unfused result followed by the mask, bit for bit."""
    from dgppo_amd import ops_nn as K_
    g = torch.Generator().manual_seed(M + N)
    X = torch.randn(M, K, generator=g).to(cuda)
    W = torch.randn(N, K, generator=g).to(cuda)                   # used transposed: the input-gradient of a Dense
    mask = torch.randn(M, N, generator=g).to(cuda)
    mask[::7] = 0.0                                               # exact zeros are masked out (y > 0)
    Y0 = torch.randn(M, N, generator=g).to(cuda)
    plain, fused = Y0.clone(), Y0.clone()
    K_.dense_fwd(X, W, None, plain, accumulate=acc, trans_w=True)
    K_.dense_fwd(X, W, None, fused, accumulate=acc, trans_w=True, relu_mask=mask)
    torch.cuda.synchronize()
    assert torch.equal(fused, torch.where(mask > 0, plain, torch.zeros_like(plain)))
    # strided mask view (leading dimension != N)
    wide = torch.randn(M, N + 5, generator=g).to(cuda)
    fused2 = Y0.clone()
    K_.dense_fwd(X, W, None, fused2, accumulate=acc, trans_w=True, relu_mask=wide[:, 2:2 + N])
    assert torch.equal(fused2, torch.where(wide[:, 2:2 + N] > 0, plain, torch.zeros_like(plain)))


def test_mean_agents_backward_with_relu_mask(cuda):
    from dgppo_amd import ops_nn as K_
    G, n, D = 50, 8, 64
    g = torch.Generator().manual_seed(4)
    dy = torch.randn(G, D, generator=g).to(cuda)
    y = torch.randn(G * n, D, generator=g).to(cuda)
    a, b = torch.empty(G * n, D, device=cuda), torch.empty(G * n, D, device=cuda)
    K_.mean_agents(dy, a, G, n, D, backward=True)
    K_.mean_agents(dy, b, G, n, D, backward=True, relu_mask=y)
    torch.cuda.synchronize()
    assert torch.equal(b, torch.where(y > 0, a, torch.zeros_like(a)))
    assert torch.equal(a.view(G, n, D)[:, 3], dy / n)


@pytest.mark.parametrize("kind,n,n_obs", [(E.LIDAR_SPREAD, 8, 3), (E.MPE_SPREAD, 3, 3)])
def test_Vh_global_info_forward_backward(cuda, kind, n, n_obs):
    """the Lagrangian baseline's constraint-value net: DecRStateFn(use_global_info=True) (value.py:61-79 — the head sees
    [x_i | mean_j x_j], 128-wide first Dense) scanned over time with its OWN carry (informarl_lagr.py:151-161)."""
    from dgppo_amd import nets, ops_nn as K_
    n_env, T_ = 3, 4
    cfg, ocfg, ag, goal, obst, hi, gr = _scene(kind, n, n_obs, n_env, T_, seed=6)
    gen = torch.Generator().manual_seed(8)
    tree = T.tree_map(lambda t: t + 0.05 * torch.randn(t.shape, generator=gen), T.init_value(3, cfg.node_dim, 2, 1, global_info=True))
    assert tuple(tree["params"]["ValueGNNHead"]["Dense_0"]["kernel"].shape) == (128, 64)
    net = nets.Net("Vhg", cfg, 1, 2, cuda)
    net.load_tree(tree)
    back = net.to_tree()
    np.testing.assert_array_equal(back["params"]["ValueGNNHead"]["Dense_0"]["kernel"], tree["params"]["ValueGNNHead"]["Dense_0"]["kernel"].numpy())
    feats = _feats(cfg, ag, goal, obst, hi, cuda)
    G = n_env * T_
    act = net.forward(feats, n_seq=n_env * n, T=T_, h0=None)       # sequences = (env, agent), zero initial carry
    lt = _leafify(tree)
    g_t = T.graph_to_torch(gr)
    sel = lambda t: {k: v.view((n_env, T_) + v.shape[1:])[:, t] for k, v in g_t.items()}
    h = torch.zeros(n_env, n, 64)
    vs = []
    for t in range(T_):
        v, h = T.value_Vh(lt, sel(t), h, n, global_info=True)
        vs.append(v)
    v_w = torch.stack(vs, 1)                                        # [n_env, T, n, 2]
    _close(act["v"].view(n_env, T_, n, 2), v_w, 1e-5, "Vh(global)")
    target = torch.randn(n_env, T_, n, 2, generator=gen)
    (0.5 * (v_w - target) ** 2).mean().backward()
    dv = torch.empty(G * n, 2, device=cuda)
    stats = torch.zeros(8, device=cuda)
    K_.value_loss(act["v"], target.reshape(-1, 2).to(cuda), dv, stats)
    net.zero_grads()
    net.backward(act, dv)
    torch.cuda.synchronize()
    _grad_tree_close(net, lt, 3e-5)


@pytest.mark.parametrize("rnn_layers,lstm", [(0, False), (2, False), (3, False), (1, True), (2, True)])
def test_policy_and_Vl_with_rnn_options(cuda, rnn_layers, lstm):
    """--no-rnn (rnn_layers = 0 here: no cell, the MLP output feeds the head, the carry passes through: policy.py:29-33)
    stacked GRU cells (--rnn-layers L: layer l consumes layer l-1's output, packed carry [h_0 | h_1 | ...],
    dgppo/nn/rnn.py:17-29) and LSTM cells (--use-lstm, packed carry [c_0 | h_0 | ...], rnn.py:22-24): chunk scan with a NON-zero initial carry, forward and backward against the oracle."""
    from dgppo_amd import nets, ops_nn as K_
    kind, n, n_obs, n_env, T_ = E.LIDAR_SPREAD, 3, 2, 3, 5
    cfg, ocfg, ag, goal, obst, hi, gr = _scene(kind, n, n_obs, n_env, T_, seed=17)
    gen = torch.Generator().manual_seed(5 + rnn_layers)
    jit = lambda tr: T.tree_map(lambda t: t + 0.05 * torch.randn(t.shape, generator=gen), tr)
    ptree = jit(T.init_policy(1, cfg.node_dim, rnn_layers=rnn_layers, lstm=lstm))
    ptree["params"]["ScaleHid"]["kernel"] = T.orthogonal(gen, 64, 64, 0.5)
    vtree = jit(T.init_value(2, cfg.node_dim, 1, 2, rnn_layers=rnn_layers, lstm=lstm))
    kw = dict(rnn="lstm" if lstm else "gru", rnn_layers=rnn_layers) if rnn_layers > 0 else dict(rnn="none", rnn_layers=0)
    CD = 64 * max(rnn_layers, 1) * (2 if lstm else 1)
    feats = _feats(cfg, ag, goal, obst, hi, cuda)
    g_t = T.graph_to_torch(gr)
    gsel = lambda t: {k: v.view((n_env, T_) + v.shape[1:])[:, t] for k, v in g_t.items()}
    G = n_env * T_
    # ---- policy
    pol = nets.Net("policy", cfg, 2, 2, cuda, **kw)
    assert pol.carry_dim == CD
    pol.load_tree(ptree)
    back = pol.to_tree()
    assert set(dict(T.tree_leaves(back))) == set(dict(T.tree_leaves(ptree)))
    h0 = 0.3 * torch.randn(n_env, n, CD, generator=gen)
    act = pol.forward(feats, n_seq=n_env * n, T=T_, h0=h0.reshape(n_env * n, CD).to(cuda))
    lt = _leafify(ptree)
    rng = torch.Generator().manual_seed(9)
    a_in = torch.tanh(torch.randn(n_env, T_, n, 2, generator=rng))
    eps_hat = torch.randn(n, 2, generator=rng)
    h = h0.clone()
    lps, ents = [], []
    for t in range(T_):
        lp, ent, h = T.policy_eval(lt, gsel(t), a_in[:, t], h, n, eps_hat)
        lps.append(lp); ents.append(ent)
    lp_w, ent_w = torch.stack(lps, 1), torch.stack(ents, 1)
    R = G * n
    lp_old = lp_w.detach() + 0.2 * torch.randn(n_env, T_, n, generator=rng)
    adv = torch.randn(n_env, T_, n, generator=rng)
    rho = torch.exp(lp_w - lp_old)
    (torch.maximum(-rho * adv, -torch.clamp(rho, 0.75, 1.25) * adv).mean() - 0.01 * ent_w.mean()).backward()
    lp = torch.empty(R, device=cuda); ent = torch.empty(R, device=cuda)
    dms = torch.empty(R, 4, device=cuda); stats = torch.zeros(8, device=cuda)
    K_.policy_head(act["ms"], eps_hat.to(cuda), a_in.reshape(R, 2).to(cuda), None, lp, ent, n, 2,
                   lp_old.reshape(R).to(cuda), adv.reshape(R).to(cuda), dms, stats, 0.25, 0.01)
    _close(lp.view(n_env, T_, n), lp_w, 1e-5, "policy log_pi")
    pol.zero_grads()
    pol.backward(act, dms)
    torch.cuda.synchronize()
    _grad_tree_close(pol, lt, 3e-5)
    # one step with hs_out: the packed carry that the rollout stores
    one = _feats(cfg, np.ascontiguousarray(ag[:, :1]), goal, obst, np.ascontiguousarray(hi[:, :1]) if hi is not None else None, cuda, tag="o")
    hs_out = torch.full((n_env * n, CD), float("nan"), device=cuda)
    pol.forward(one, n_seq=n_env * n, T=1, h0=h0.reshape(n_env * n, CD).to(cuda), hs_out=hs_out, train=False, tag="o")
    with torch.no_grad():
        _, h1 = T.policy_mode(ptree, gsel(0), h0, n)
    _close(hs_out.view(n_env, n, CD), h1, 1e-5, "carry after one step")
    # ---- Vl (one sequence per env, n_inner = 1)
    vl = nets.Net("Vl", cfg, 2, 1, cuda, **kw)
    vl.load_tree(vtree)
    act = vl.forward(feats, n_seq=n_env, T=T_, h0=None)
    lv = _leafify(vtree)
    h = torch.zeros(n_env, 1, CD)
    vs = []
    for t in range(T_):
        v, h = T.value_Vl(lv, gsel(t), h, n)
        vs.append(v)
    v_w = torch.stack(vs, 1)
    _close(act["v"].view(n_env, T_), v_w, 1e-5, "Vl")
    target = torch.randn(n_env, T_, generator=gen)
    (0.5 * (v_w - target) ** 2).mean().backward()
    dv = torch.empty(G, 1, device=cuda)
    K_.value_loss(act["v"], target.reshape(-1, 1).to(cuda), dv, torch.zeros(8, device=cuda))
    vl.zero_grads()
    vl.backward(act, dv)
    torch.cuda.synchronize()
    _grad_tree_close(vl, lv, 3e-5)


@pytest.mark.parametrize("M,K,N,trans", [(131072, 8, 24, False), (40000, 4, 64, True), (9999, 1, 64, True), (5000, 2, 64, True),
                                         (777, 16, 192, False), (3000, 7, 10, False), (1, 8, 24, False), (4097, 3, 5, True)])
def test_dense_small_k_path(cuda, M, K, N, trans):
    """K <= 16 takes dense_smallk_kernel (W in LDS, a thread per four output columns): plain, bias + relu, accumulate into a
    strided view, and the ReLU-mask epilogue, against torch fp32 — and against the tiled kernel (DGPPO_DENSE_NO_SMALLK)."""
    from dgppo_amd import ops_nn as K_
    g = torch.Generator().manual_seed(M + 31 * K + N)
    X = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) if trans else torch.randn(K, N, generator=g)
    b = torch.randn(N, generator=g)
    Wm = W.T if trans else W
    Y = torch.full((M, N), float("nan"), device=cuda)
    K_.dense_fwd(X.to(cuda), W.to(cuda), b.to(cuda), Y, act=1, trans_w=trans)
    _close(Y, torch.relu(X @ Wm + b), 2e-6 * math.sqrt(K) + 1e-6, "small-k relu")
    Y0 = torch.randn(M, N + 6, generator=g)
    mask = torch.randn(M, N, generator=g)
    mask[::5] = 0.0
    Yw = Y0.to(cuda).clone()
    K_.dense_fwd(X.to(cuda), W.to(cuda), None, Yw[:, 2:2 + N], accumulate=True, trans_w=trans, relu_mask=mask.to(cuda))
    want = torch.where(mask > 0, Y0[:, 2:2 + N] + X @ Wm, torch.zeros(M, N))
    _close(Yw[:, 2:2 + N], want, 2e-6 * math.sqrt(K) + 1e-6, "small-k acc+mask")
    assert torch.equal(Yw[:, :2].cpu(), Y0[:, :2]) and torch.equal(Yw[:, 2 + N:].cpu(), Y0[:, 2 + N:])


@pytest.mark.parametrize("M,K,N,bias", [(131072, 8, 24, True), (4096, 16, 64, True), (10001, 3, 8, False), (50000, 8, 24, False)])
def test_dense_bwd_w_small_k_path(cuda, M, K, N, bias):
    """K <= 16, N <= 64, N % 4 == 0, M >= 4096: dense_bwd_w_smallk_kernel (register accumulators, LDS atomics per workgroup,
    common second-stage reduction) against float64."""
    from dgppo_amd import ops_nn as K_
    g = torch.Generator().manual_seed(M + K + N)
    X = torch.randn(M, K, generator=g)
    dY = torch.randn(M, N, generator=g)
    dW0 = torch.randn(K, N, generator=g)
    db0 = torch.randn(N, generator=g)
    dW, db = dW0.to(cuda).clone(), (db0.to(cuda).clone() if bias else None)
    K_.dense_bwd_w(X.to(cuda), dY.to(cuda), dW, db)
    _close(dW, dW0.double() + X.double().T @ dY.double(), 3e-6 * math.sqrt(M), "dW small-k")
    if bias:
        _close(db, db0.double() + dY.double().sum(0), 3e-6 * math.sqrt(M), "db small-k")


def test_dense_bwd_w_deferred_batch_equals_immediate(cuda):
    """dgppo_dense_bwd_w_deferred + dgppo_dense_bwd_w_reduce_batch (ops_nn.BwdWBatch: one reduce launch for many weight
    gradients, each with its own workspace region) against the immediate two-stage call: same sums, accumulation into a
    shared dW from two calls included, the small-K and the tiny (<= 4 workgroups, atomic) paths mixed in."""
    from dgppo_amd import nets, ops_nn as K_
    g = torch.Generator().manual_seed(3)
    shapes = [(20000, 64, 64, True), (20000, 64, 192, True), (9000, 144, 64, False), (20000, 8, 24, True), (100, 32, 16, True),
              (20000, 64, 4, True)] * 3                                                      # 18 gradients: two reduce launches
    data = [(torch.randn(M, Kd, generator=g).to(cuda), torch.randn(M, Nd, generator=g).to(cuda), b) for M, Kd, Nd, b in shapes]
    arena = nets.Arena(cuda)
    def run(batched, warm=False):
        outs = [(torch.ones(X.shape[1], dY.shape[1], device=cuda), torch.ones(dY.shape[1], device=cuda) if b else None) for X, dY, b in data]
        shared = torch.zeros(64, 64, device=cuda)                                            # two calls accumulate into it
        big = max([0] + [v.numel() for v in arena.bufs.values()])
        ctx = K_.BwdWBatch(cuda, lambda nf: arena.get("ws", max(nf, big))) if batched else None
        if ctx is not None:
            ctx.__enter__()
        for (X, dY, b), (dW, db) in zip(data, outs):
            K_.dense_bwd_w(X, dY, dW, db)
        K_.dense_bwd_w(data[0][0], data[0][1], shared)
        K_.dense_bwd_w(data[0][0], data[0][1], shared)
        if ctx is not None:
            if warm:                             # cold: the workspace grows (and flushes) on the way; warm: one batch at the end
                assert len(ctx.descs) >= 16
            ctx.__exit__(None, None, None)
        torch.cuda.synchronize()
        return outs, shared
    a, sa = run(False)
    run(True)                                    # cold pass sizes the workspace
    b, sb = run(True, warm=True)
    for (dWa, dba), (dWb, dbb), (X, dY, hb) in zip(a, b, data):
        scale = float(dWa.abs().max())
        assert float((dWa - dWb).abs().max()) <= 2e-6 * scale * math.sqrt(X.shape[0]) / 100 + 1e-5 * scale
        if hb:
            assert float((dba - dbb).abs().max()) <= 1e-5 * float(dba.abs().max())
    assert float((sa - sb).abs().max()) <= 1e-5 * float(sa.abs().max())
    want = 2.0 * (data[0][0].double().T @ data[0][1].double())
    assert float((sb.double() - want).abs().max()) <= 3e-6 * math.sqrt(20000) * float(want.abs().max())


@pytest.mark.parametrize("kind,n,n_obs,G", [(E.LIDAR_SPREAD, 8, 3, 4096), (E.MPE_SPREAD, 5, 3, 512), (E.LIDAR_TARGET, 4, 2, 512)])
def test_device_networks_are_agent_permutation_equivariant(cuda, kind, n, n_obs, G):
    """SURVEY §8(c)(3) on the HIP path itself, at a size the oracle could not walk (4096 graphs of the benchmark topology):
    renumbering the agents (their states, their LiDAR hits and — for the Target tasks — their goals) permutes the actor's
    outputs and the per-agent constraint values the same way and leaves the pooled cost value unchanged.  fp32: the
    attention sums run over the same slots in a different order (1e-5)."""
    from dgppo_amd import nets, init
    cfg, ocfg, ag, goal, obst, hi, _ = _scene(kind, n, n_obs, 64, 1, seed=11)
    rep = G // 64
    rng = np.random.default_rng(5)
    jit = lambda a: None if a is None else (np.repeat(a, rep, axis=0) + rng.normal(scale=0.02, size=(a.shape[0] * rep,) + a.shape[1:])).astype(np.float32)
    ag, hi = jit(ag), jit(hi)
    goal = np.repeat(goal, rep, axis=0)
    obst = None if obst is None else np.repeat(obst, rep, axis=0)
    perm = rng.permutation(n)
    ag_p = np.ascontiguousarray(ag[:, :, perm])
    hi_p = None if hi is None else np.ascontiguousarray(hi[:, :, perm])
    goal_p = goal if ocfg.is_spread else np.ascontiguousarray(goal[:, perm])   # Target: goal i belongs to agent i and moves with it
    trees = {"policy": init.init_policy(3, cfg.node_dim, 2, 2), "Vl": init.init_value(4, cfg.node_dim, 1, 2, 2),
             "Vh": init.init_value(5, cfg.node_dim, 2, 1, 3)}
    gen = np.random.default_rng(9)
    for kindname, n_out, layers in (("policy", 2, 2), ("Vl", 1, 2), ("Vh", 2, 1)):
        tree = T.tree_map(lambda a: torch.from_numpy(a + 0.05 * gen.standard_normal(a.shape).astype(np.float32)), trees[kindname])
        net = nets.Net(kindname, cfg, layers, n_out, cuda)
        net.load_tree(tree)
        outs = []
        for a_, h_, g_, tag in ((ag, hi, goal, "a"), (ag_p, hi_p, goal_p, "b")):
            feats = _feats(cfg, a_, g_, obst, h_, cuda, tag=kindname + tag)
            act = net.forward(feats, n_seq=G * (1 if kindname == "Vl" else n), T=1, h0=None, tag="f" + tag, train=False)
            torch.cuda.synchronize()
            outs.append((act["ms"] if kindname == "policy" else act["v"]).clone())
        a0, a1 = outs
        if kindname == "Vl":
            _close(a1, a0, 1e-5, "Vl is invariant")
        else:
            w = a0.shape[-1]
            _close(a1.view(G, n, w), a0.view(G, n, w)[:, perm], 1e-5, f"{kindname} is equivariant")
            assert float((a0.view(G, n, w)[:, perm] - a0.view(G, n, w)).abs().max()) > 1e-3      # the permutation matters


def test_graph_feats_rejects_permuted_views(cuda):
    """dgppo_graph_feats addresses its records as data_ptr + env * stride + time * stride: a view whose innermost record is
    not dense (e.g. agents permuted by fancy indexing on the host and moved over as a strided tensor) must be refused, not
    read as if it were dense."""
    from dgppo_amd import nets
    cfg, ocfg, ag, goal, obst, hi, _ = _scene(E.LIDAR_SPREAD, 4, 2, 3, 2, seed=1)
    arena = nets.Arena(cuda)
    f = nets.GraphFeats(cfg, 6, arena, "x")
    agd = torch.from_numpy(ag).to(cuda)
    hid = torch.from_numpy(hi).to(cuda)
    bad = agd.transpose(2, 3)                                    # [env, T, sd, n] storage seen as [.., n, sd]: same numel, wrong layout
    args = (torch.from_numpy(goal).to(cuda), torch.from_numpy(obst).to(cuda))
    n, sd, k = 4, 4, cfg.top_k
    with pytest.raises(ValueError, match="dense"):
        f.compute(bad, 2 * n * sd, n * sd, *args, hid, 2 * n * k * 2, n * k * 2, None, 3, 2)
    f8 = nets.GraphFeats(cfg, 8, arena, "y")
    with pytest.raises(ValueError, match="beyond its storage"):
        f8.compute(agd, 2 * n * sd, n * sd, *args, hid, 2 * n * k * 2, n * k * 2, None, 4, 2)       # one env too many
    f.compute(agd, 2 * n * sd, n * sd, *args, hid, 2 * n * k * 2, n * k * 2, None, 3, 2)            # the dense record is fine
