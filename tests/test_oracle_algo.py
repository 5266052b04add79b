"""Closed-form checks that pin oracle/algo_ref.py (CPU only)."""
import numpy as np

from oracle import algo_ref as A


def _rand(T=12, n=3, nh=2, seed=0):
    r = np.random.default_rng(seed)
    return (r.normal(size=(T, n, nh)).astype(np.float32), r.normal(size=T).astype(np.float32),
            r.normal(size=(T + 1, n, nh)).astype(np.float32), r.normal(size=T + 1).astype(np.float32))


def test_Ql_equals_textbook_gae_plus_V():
    hs, l, Vh, Vl = _rand()
    g, lam = 0.99, 0.95
    Qh, Ql = A.compute_dec_ocp_gae(hs, l, Vh, Vl, g, lam)
    T = len(l)
    adv = np.zeros(T)
    nxt = 0.0
    for t in range(T - 1, -1, -1):
        delta = l[t] + g * Vl[t + 1] - Vl[t]
        nxt = delta + g * lam * nxt
        adv[t] = nxt
    np.testing.assert_allclose(Ql, adv + Vl[:-1], rtol=2e-5, atol=2e-5)


def test_lambda_limits():
    hs, l, Vh, Vl = _rand(T=6)
    g = 0.9
    Qh0, Ql0 = A.compute_dec_ocp_gae(hs, l, Vh, Vl, g, 0.0)       # lambda = 0: one-step bootstrap
    np.testing.assert_allclose(Ql0, l + g * Vl[1:], rtol=1e-5, atol=1e-6)
    hm = hs.max(-1, keepdims=True)
    np.testing.assert_allclose(Qh0, np.maximum(hs, (1 - g) * hm + g * Vh[1:]), rtol=1e-5, atol=1e-6)
    Qh1, Ql1 = A.compute_dec_ocp_gae(hs, l, Vh, Vl, g, 1.0)       # lambda = 1: Monte-Carlo to the horizon
    ret = Vl[-1]
    rh = Vh[-1]
    for t in range(len(l) - 1, -1, -1):
        ret = l[t] + g * ret
        rh = np.maximum(hs[t], (1 - g) * hs[t].max(-1, keepdims=True) + g * rh)
        np.testing.assert_allclose(Ql1[t], ret, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(Qh1[t], rh, rtol=1e-5, atol=1e-6)


def test_constant_costs_fixed_point():
    T, n, nh = 8, 2, 2
    hs = np.full((T, n, nh), 0.3, np.float32)
    Vh = np.full((T + 1, n, nh), 0.3, np.float32)
    Qh, _ = A.compute_dec_ocp_gae(hs, np.zeros(T, np.float32), Vh, np.zeros(T + 1, np.float32), 0.99, 0.95)
    np.testing.assert_allclose(Qh, 0.3, rtol=1e-5)     # max(h, (1-g) h + g h) = h, lambda weights sum to 1


def test_advantage_block():
    r = np.random.default_rng(1)
    B, T, n, nh = 3, 10, 2, 2
    Ql = r.normal(size=(B, T)).astype(np.float32)
    Vl = r.normal(size=(B, T + 1)).astype(np.float32)
    Vh = (r.normal(size=(B, T + 1, n, nh)) * 0.02 - 0.03).astype(np.float32)
    Aout, safe = A.advantage(Ql, Vl, Vh, 0.03, 10.0, 1e-2, 1.0)
    Al = Ql - Vl[:, :-1]
    Al = (Al - Al.mean(1, keepdims=True)) / (Al.std(1, keepdims=True) + 1e-8)
    deriv = (Vh[:, 1:] - Vh[:, :-1]) / 0.03 + 10 * Vh[:, :-1]
    assert 0 < safe < 1
    for b, t, a in [(0, 0, 0), (1, 5, 1), (2, 9, 0)]:
        s = np.all(deriv[b, t, a] <= 0)
        want = -((Al[b, t] if s else 0.0) + max(np.maximum(deriv[b, t, a] + 1e-2, 0)))
        np.testing.assert_allclose(Aout[b, t, a], want, rtol=1e-5, atol=1e-6)
    assert A.cbf_weight_schedule(1.0, 49, 100) == 1.0 and A.cbf_weight_schedule(1.0, 50, 100) == 2.0
    assert A.cbf_weight_schedule(1.0, 75, 100) == 4.0


def test_adam_three_steps_hand_calculation():
    p, m, v, c = np.array([1.0]), np.zeros(1), np.zeros(1), 0
    lr = 0.1
    # max_norm large: no clipping; Adam with constant gradient moves by lr each step (m_hat = g, sqrt(v_hat) = |g|)
    for k in range(3):
        p, m, v, c, norm, bad = A.clip_adam(p, np.array([0.5]), m, v, c, lr, 1e9)
        np.testing.assert_allclose(p, 1.0 - lr * (k + 1), rtol=1e-6)
    assert c == 3 and not bad and abs(norm - 0.5) < 1e-12
    # clipping: g = [3, 4] (norm 5) with max_norm 2 -> scaled by 2/5
    p2, m2, v2, c2, norm, _ = A.clip_adam(np.zeros(2), np.array([3.0, 4.0]), np.zeros(2), np.zeros(2), 0, lr, 2.0)
    np.testing.assert_allclose(m2, 0.1 * np.array([1.2, 1.6]))
    assert norm == 5.0
    # non-finite gradient: apply_if_finite skips the step and leaves the inner state untouched
    p3, m3, v3, c3, _, bad = A.clip_adam(p2, np.array([np.nan, 1.0]), m2, v2, c2, lr, 2.0)
    assert bad and c3 == c2 and np.array_equal(p3, p2) and np.array_equal(m3, m2)


def test_informarl_targets_known_answers():
    """InforMARL targets (informarl.py:323-336).  (1) the shaped stage cost; (2) with costs far below zero the constraint
    never binds and Ql is the textbook GAE(lambda) return of l; (3) the advantage is -(Ql - Vl) standardised per env
    (population std) and broadcast over agents; (4) the x5 / x25 cost-weight schedule."""
    from oracle import algo_ref as A
    rng = np.random.default_rng(3)
    B, T, n, nh = 3, 12, 4, 2
    gamma, lam, w = 0.99, 0.95, 0.7
    rewards = rng.normal(size=(B, T)).astype(np.float32)
    Vl = rng.normal(size=(B, T + 1)).astype(np.float32)
    costs = rng.normal(size=(B, T, n, nh)).astype(np.float32) - 50.0          # never binding
    Ql0, adv0 = A.informarl_targets(costs, rewards, Vl, gamma, lam, w)
    # all costs negative: max(cost, 0) = 0, l = -reward; textbook recursion A_t = delta_t + gamma*lam*A_{t+1}
    l = -rewards
    want = np.zeros((B, T))
    for b in range(B):
        a_next = 0.0
        for t in reversed(range(T)):
            delta = l[b, t] + gamma * Vl[b, t + 1] - Vl[b, t]
            a_next = delta + gamma * lam * a_next
            want[b, t] = a_next + Vl[b, t]
    np.testing.assert_allclose(Ql0, want, rtol=2e-5, atol=2e-5)
    Al = want - Vl[:, :-1]
    Aw = -(Al - Al.mean(1, keepdims=True)) / (Al.std(1, keepdims=True) + 1e-8)
    np.testing.assert_allclose(adv0, np.repeat(Aw[:, :, None], n, -1), rtol=2e-4, atol=2e-4)
    assert adv0.shape == (B, T, n) and np.allclose(adv0[..., 0], adv0[..., 3])
    # positive costs enter the stage cost with weight w: shifting one (t, agent, component) by +c raises l_t by w*c only
    costs2 = costs.copy()
    costs2[:, 5, 2, 1] = 3.0                                                  # still <= the running max structure? check l only
    l2 = -rewards + w * np.maximum(costs2, 0).sum(-1).sum(-1)
    assert np.allclose(l2[:, 5] - l[:, 5], w * 3.0, atol=1e-6) and np.allclose(np.delete(l2, 5, 1), np.delete(l, 5, 1))
    # schedule
    assert A.cost_weight_schedule(0.5, 10, 100, enabled=False) == 0.5
    assert [A.cost_weight_schedule(0.5, s, 100, enabled=True) for s in (0, 49, 50, 74, 75, 99)] == [0.5, 0.5, 2.5, 2.5, 12.5, 12.5]


def test_lagrangian_advantage_and_multiplier_known_answers():
    """informarl_lagr.py:219-235,300-308 with hand-checkable numbers."""
    from oracle import algo_ref as A
    # one env, T = 2, one agent, one component: Al = Ql - Vl = [1, 3] -> standardised [-1, 1] -> negated [1, -1];
    # Ah = Qh - Vh = [2, 0] -> [1, -1]; lagr = 0.5  ->  A = [1, -1] - 0.5 * [1, -1] = [0.5, -0.5]
    Ql = np.array([[2.0, 5.0]], np.float32); Vl = np.array([[1.0, 2.0, 9.0]], np.float32)
    Qh = np.array([[[[3.0]], [[1.0]]]], np.float32); Vh = np.array([[[[1.0]], [[1.0]], [[7.0]]]], np.float32)
    A_, Ah = A.advantage_lagr(Ql, Vl, Qh, Vh, np.array([[0.5]], np.float32))
    np.testing.assert_allclose(Ah[0, :, 0, 0], [1.0, -1.0], atol=1e-6)
    np.testing.assert_allclose(A_[0, :, 0], [0.5, -0.5], atol=1e-6)
    # two components average (mean over h): lagr = [1, 3], Ah identical in both -> A = -Al - (1 + 3) / 2 * Ah
    Qh2 = np.repeat(Qh, 2, axis=-1); Vh2 = np.repeat(Vh, 2, axis=-1)
    A2, _ = A.advantage_lagr(Ql, Vl, Qh2, Vh2, np.array([[1.0, 3.0]], np.float32))
    np.testing.assert_allclose(A2[0, :, 0], [1.0 - 2.0, -1.0 + 2.0], atol=1e-6)
    # multiplier: rho = 1 (lp_new == lp_old): delta = -mean(Vh (1 - gamma) + Ah); lagr' = relu(lagr - delta lr)
    lp = np.zeros((1, 2, 1), np.float32)
    Vh_T = np.array([[[[2.0]], [[4.0]]]], np.float32); Ah_T = np.array([[[[1.0]], [[-3.0]]]], np.float32)
    got = A.lagr_update(np.array([[0.5]], np.float32), lp, lp, Vh_T, Ah_T, gamma=0.9, lr=0.1)
    want = 0.5 + 0.1 * ((2.0 * 0.1 + 1.0) + (4.0 * 0.1 - 3.0)) / 2
    np.testing.assert_allclose(got, [[want]], atol=1e-6)
    # clipped at zero, and the ratio scales the advantage term
    got = A.lagr_update(np.array([[0.01]], np.float32), lp + np.log(2.0).astype(np.float32), lp, Vh_T * 0, -np.abs(Ah_T), 0.9, 1.0)
    assert got[0, 0] == 0.0
