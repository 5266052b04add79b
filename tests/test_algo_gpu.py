"""GPU parity of GAE / advantage / clip+Adam (HIP via the C ABI) against oracle/algo_ref.py."""
import numpy as np
import pytest
import torch

from oracle import algo_ref as A

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,T,n,nh,lam", [(7, 128, 8, 2, 0.95), (3, 16, 3, 2, 0.95), (2, 128, 16, 2, 0.95), (4, 33, 1, 3, 0.95),
                                              # column-parallel kernel: every (rows per lane, lanes per column) instantiation, ragged
                                              # last chunk, lambda at its 0.5 limit and at 1; below 0.5 the row-parallel kernels
                                              (5, 32, 4, 3, 0.5), (3, 65, 2, 2, 1.0), (2, 200, 3, 2, 0.9), (300, 128, 8, 2, 0.97),
                                              (3, 40, 3, 2, 0.3), (2, 7, 2, 2, 0.0)])
def test_gae(cuda, monkeypatch, B, T, n, nh, lam):
    from dgppo_amd import ops_algo as O
    r = np.random.default_rng(B + T)
    costs = r.uniform(-1, 1, size=(B, T, n, nh)).astype(np.float32)
    rew = (-r.uniform(0, 0.02, size=(B, T))).astype(np.float32)
    Vh = r.uniform(-1, 1, size=(B, T + 1, n, nh)).astype(np.float32)
    Vl = r.uniform(0, 1, size=(B, T + 1)).astype(np.float32)
    Qh_w, Ql_w = A.gae_batch(costs[:8], rew[:8], Vh[:8], Vl[:8], 0.99, lam)       # the oracle is a Python loop: 8 envs of it
    d = lambda x: torch.from_numpy(x).to(cuda)
    outs = {}
    for name, env in (("cols", {}), ("rows", {"DGPPO_GAE_ROWS": "1"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        Qh = torch.full((B, T, n, nh), float("nan"), device=cuda)
        Ql = torch.full((B, T), float("nan"), device=cuda)
        O.gae(d(costs), d(rew), d(Vh), d(Vl), O.lam_pow_table(lam, T, cuda), 0.99, lam, Qh, Ql)
        outs[name] = (Qh.cpu().numpy(), Ql.cpu().numpy())
        for k in env:
            monkeypatch.delenv(k)
        np.testing.assert_allclose(outs[name][0][:8], Qh_w, rtol=0, atol=1e-5, err_msg=name)
        np.testing.assert_allclose(outs[name][1][:8], Ql_w, rtol=0, atol=1e-5, err_msg=name)
    # the two kernel families agree on every env (the oracle covers the first 8)
    np.testing.assert_allclose(outs["cols"][0], outs["rows"][0], rtol=0, atol=1e-5)
    np.testing.assert_allclose(outs["cols"][1], outs["rows"][1], rtol=0, atol=1e-5)


@pytest.mark.parametrize("family", ["cols", "rows", "generic"])
def test_gae_propagates_nan_like_the_reference(cuda, monkeypatch, family):
    """jnp.maximum / .max(-1) propagate NaN (algo/utils.py:39-44) and `mask * NaN` is NaN; v_max_f32 drops a NaN operand.
    A NaN cost (a NaN LiDAR hit point reaches get_cost), a NaN reward and a NaN value must mark exactly the entries the
    oracle marks: Qh[t' <= t, agent, all components] for a cost, Ql[t' <= t] for a reward, t' < t for an inserted value —
    and nothing else.  All three kernel families (column-parallel default, row-parallel, generic)."""
    from dgppo_amd import ops_algo as O
    B, T, n, nh = 6, (300 if family == "generic" else 40), 3, 2
    lam = 0.95
    if family == "rows":
        monkeypatch.setenv("DGPPO_GAE_ROWS", "1")
    r = np.random.default_rng(3)
    costs = r.uniform(-1, 1, size=(B, T, n, nh)).astype(np.float32)
    rew = (-r.uniform(0, 0.02, size=(B, T))).astype(np.float32)
    Vh = r.uniform(-1, 1, size=(B, T + 1, n, nh)).astype(np.float32)
    Vl = r.uniform(0, 1, size=(B, T + 1)).astype(np.float32)
    costs[0, 25, 1, 0] = np.nan                       # one component of one agent in the middle
    costs[1, T - 1, 2, 1] = np.nan                    # last step: every Qh of that agent
    costs[2, 0, 0, 1] = np.nan                        # first step: only Qh[0]
    rew[3, 17] = np.nan                               # the cost-value column
    Vh[4, 30, 1, 1] = np.nan                          # an inserted value row: t' < 30 of that (agent, component) only
    Qh_w, Ql_w = A.gae_batch(costs, rew, Vh, Vl, 0.99, lam)
    assert np.isnan(Qh_w[0, :26, 1, :]).all() and not np.isnan(Qh_w[0, 26:]).any() and not np.isnan(Qh_w[0, :, [0, 2]]).any()
    assert np.isnan(Qh_w[4, :30, 1, 1]).all() and not np.isnan(Qh_w[4, 30:, 1, 1]).any() and not np.isnan(Qh_w[4, :, 1, 0]).any()
    assert np.isnan(Ql_w[3, :18]).all() and not np.isnan(Ql_w[3, 18:]).any() and not np.isnan(Qh_w[3]).any()
    d = lambda x: torch.from_numpy(x).to(cuda)
    Qh = torch.zeros(B, T, n, nh, device=cuda); Ql = torch.zeros(B, T, device=cuda)
    O.gae(d(costs), d(rew), d(Vh), d(Vl), O.lam_pow_table(lam, T, cuda), 0.99, lam, Qh, Ql)
    Qh, Ql = Qh.cpu().numpy(), Ql.cpu().numpy()
    assert np.array_equal(np.isnan(Qh), np.isnan(Qh_w)), "NaN pattern of Qh differs from the oracle's"
    assert np.array_equal(np.isnan(Ql), np.isnan(Ql_w)), "NaN pattern of Ql differs from the oracle's"
    np.testing.assert_allclose(np.nan_to_num(Qh), np.nan_to_num(Qh_w), rtol=0, atol=1e-5)
    np.testing.assert_allclose(np.nan_to_num(Ql), np.nan_to_num(Ql_w), rtol=0, atol=1e-5)


def test_advantage(cuda):
    from dgppo_amd import ops_algo as O
    r = np.random.default_rng(5)
    B, T, n, nh = 9, 128, 8, 2
    Ql = r.normal(size=(B, T)).astype(np.float32)
    Vl = r.normal(size=(B, T + 1)).astype(np.float32)
    Vh = (r.normal(size=(B, T + 1, n, nh)) * 0.02 - 0.03).astype(np.float32)
    want, safe = A.advantage(Ql, Vl, Vh, 0.03, 10.0, 1e-2, 2.0)
    d = lambda x: torch.from_numpy(x).to(cuda)
    adv = torch.empty(B, T, n, device=cuda)
    stats = torch.zeros(8, device=cuda)
    O.advantage(d(Ql), d(Vl), d(Vh), 0.03, 10.0, 1e-2, 2.0, adv, stats)
    got = adv.cpu().numpy()
    # The safe gate is a hard threshold on the fp32 quantity cdot = (Vh[t+1]-Vh[t])/dt + alpha*Vh[t] (dgppo.py:246-251).
    # Gate flips and numeric error are counted SEPARATELY: a flip is legitimate only where |cdot| is at rounding level
    # (the kernel and numpy may round cdot differently by an ulp); every other entry must agree to 1e-5.
    deriv = (Vh[:, 1:] - Vh[:, :-1]) / np.float32(0.03) + np.float32(10.0) * Vh[:, :-1]
    rounding_level = (np.abs(deriv) < 1e-5).any(axis=-1)
    bad = np.abs(got - want) > 1e-5 * np.maximum(1, np.abs(want))
    assert not (bad & ~rounding_level).any(), "numeric error above 1e-5 away from the gate threshold"
    assert (bad & rounding_level).sum() <= 4, "gate flips at the threshold should be a handful at most"
    assert abs(stats[0].item() / (B * T * n) - safe) < 1e-3
    assert 0.05 < safe < 0.95


def test_clip_adam_matches_optax_semantics(cuda):
    from dgppo_amd import ops_algo as O
    r = np.random.default_rng(0)
    n = 62660
    p = r.normal(size=n).astype(np.float32)
    pd = torch.from_numpy(p.copy()).to(cuda)
    m = torch.zeros(n, device=cuda); v = torch.zeros(n, device=cuda)
    from dgppo_amd import _native as N
    st = torch.zeros(N.OPT_STATE_FLOATS, device=cuda)
    pr, mr, vr, cr = p.astype(np.float64), np.zeros(n), np.zeros(n), 0
    for k in range(4):
        g = (r.normal(size=n) * (0.001 if k == 1 else 0.05)).astype(np.float32)     # k=1: below max_norm -> no clipping
        if k == 2:
            g[123] = np.inf                                                            # skipped step
        O.clip_adam_step(pd, torch.from_numpy(g).to(cuda), m, v, st, 3e-4, 2.0)
        pr, mr, vr, cr, norm, bad = A.clip_adam(pr, g, mr, vr, cr, 3e-4, 2.0)
        s = st.cpu().numpy()
        assert s[5] == float(bad) and s[2] == cr and s[3] == k + 1
        if not bad:
            np.testing.assert_allclose(s[4], norm, rtol=1e-5)
        np.testing.assert_allclose(pd.cpu().numpy(), pr, rtol=0, atol=2e-6)
    np.testing.assert_allclose(m.cpu().numpy(), mr, rtol=1e-4, atol=1e-8)
    np.testing.assert_allclose(v.cpu().numpy(), vr, rtol=1e-4, atol=1e-10)


def test_lagrangian_kernels(cuda):
    """dgppo_advantage_lagr / dgppo_lagr_update / dgppo_relu_fwd against oracle/algo_ref.py (informarl_lagr.py:213-235,286-309)"""
    from dgppo_amd import ops_algo as O
    r = np.random.default_rng(3)
    B, T, n, nh = 7, 32, 8, 2
    Ql = r.normal(size=(B, T)).astype(np.float32); Vl = r.normal(size=(B, T + 1)).astype(np.float32)
    Qh = r.normal(size=(B, T, n, nh)).astype(np.float32); Vh = r.normal(size=(B, T + 1, n, nh)).astype(np.float32)
    lagr = r.uniform(0, 1, size=(n, nh)).astype(np.float32)
    d = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(cuda)
    adv = torch.empty(B, T, n, device=cuda); Ah = torch.empty(B, T, n, nh, device=cuda)
    O.advantage_lagr(d(Ql), d(Vl), d(Qh), d(Vh), d(lagr), adv, Ah)
    wA, wAh = A.advantage_lagr(Ql, Vl, Qh, Vh, lagr)
    np.testing.assert_allclose(Ah.cpu().numpy(), wAh, atol=1e-5)
    np.testing.assert_allclose(adv.cpu().numpy(), wA, atol=1e-5)
    lp_new = (r.normal(size=(B, T, n)) * 0.3).astype(np.float32); lp_old = (r.normal(size=(B, T, n)) * 0.3).astype(np.float32)
    lg = d(lagr.copy()); sums = torch.zeros(n * nh, device=cuda)
    for lr in (0.5, 50.0):                                       # the second step drives some multipliers to the clip at 0
        O.lagr_update(d(lp_new), d(lp_old), d(Vh), d(wAh), lg, sums, 0.99, lr)
        lagr = A.lagr_update(lagr, lp_new, lp_old, Vh[:, :T], wAh, 0.99, lr)
        np.testing.assert_allclose(lg.cpu().numpy(), lagr, atol=1e-5 * max(1.0, float(np.abs(lagr).max())))
        assert float(sums.abs().max()) == 0.0
    assert (lagr == 0).any() and (lagr > 0).any()
    x = d(r.normal(size=(1000,)).astype(np.float32)); y = torch.empty_like(x)
    O.relu_fwd(x, y)
    assert torch.equal(y, torch.clamp_min(x, 0.0))


def test_gae_full_size_is_positively_homogeneous(cuda):
    """A size-independent property at the benchmark size (B = 4096 envs, T = 128, n = 8, 2 costs), where the Python oracle
    cannot go: the Dec-OCP recursion is built from +, max and non-negative weights, so scaling costs, rewards and values by
    2 scales Qh and Ql by 2 — exactly in fp32 (a power of two); every Qh lies between the extremes of the costs / values it
    is a convex combination of; and with Vh = costs = 0 the constraint targets are exactly 0."""
    from dgppo_amd import ops_algo as O
    B, T, n, nh = 4096, 128, 8, 2
    g = torch.Generator().manual_seed(1)
    costs = (torch.rand(B, T, n, nh, generator=g) * 2 - 1).to(cuda)
    rew = (-torch.rand(B, T, generator=g) * 0.02).to(cuda)
    Vh = (torch.rand(B, T + 1, n, nh, generator=g) * 2 - 1).to(cuda)
    Vl = torch.rand(B, T + 1, generator=g).to(cuda)
    lp = O.lam_pow_table(0.95, T, cuda)

    def run(c, r, vh, vl):
        Qh = torch.empty(B, T, n, nh, device=cuda); Ql = torch.empty(B, T, device=cuda)
        O.gae(c, r, vh, vl, lp, 0.99, 0.95, Qh, Ql)
        return Qh, Ql
    Qh1, Ql1 = run(costs, rew, Vh, Vl)
    Qh2, Ql2 = run(2 * costs, 2 * rew, 2 * Vh, 2 * Vl)
    torch.cuda.synchronize()
    assert torch.equal(Qh2, 2 * Qh1) and torch.equal(Ql2, 2 * Ql1)
    assert torch.isfinite(Qh1).all() and torch.isfinite(Ql1).all()
    # bounds: every Qh is a convex combination of max-discounted values, so it lies between the extremes of its inputs
    lo = torch.minimum(costs.amin(dim=1), Vh.amin(dim=1)).unsqueeze(1) - 1e-5
    hi = torch.maximum(costs.amax(dim=1), Vh.amax(dim=1)).unsqueeze(1) + 1e-5
    assert ((Qh1 >= lo) & (Qh1 <= hi)).all()
    Qh0, _ = run(torch.zeros_like(costs), rew, torch.zeros_like(Vh), Vl)
    assert float(Qh0.abs().max()) == 0.0
