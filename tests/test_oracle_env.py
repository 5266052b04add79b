"""Known-answer and property tests that pin oracle/env_np.py (CPU only).

The reference is not importable here and ships no golden vectors, so these analytic cases are what
pins the oracle ("parity unpinned" w.r.t. outputs of the reference itself)."""
import numpy as np
import pytest

from oracle import env_np as E

f32 = np.float32


def test_philox_known_answers():
    # Random123 kat_vectors for philox4x32-10
    assert E.philox4x32((0, 0, 0, 0), (0, 0)) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert E.philox4x32((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert E.philox4x32((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0)) == \
        [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def _square(cx, cy, side, theta=0.0):
    return E.make_rect(np.array([cx, cy], f32), f32(side), f32(side), f32(theta))


def test_ray_hits_axis_aligned_square_closed_form():
    cfg = E.EnvCfg(E.LIDAR_SPREAD, n_agents=1, n_obs=1, n_rays=4, top_k=2)
    # agent at (0.5, 0.5); square of side 0.2 centred at (0.8, 0.6); the ray at +30 deg hits its left face x = 0.7
    obst = _square(0.8, 0.6, 0.2)[None, None]
    pos = np.array([[[0.5, 0.5]]], f32)
    a30 = np.deg2rad(30.0)
    rc = np.array([np.cos(a30 + np.pi), np.cos(a30 - np.pi / 2), np.cos(a30), np.cos(a30 + np.pi / 2)], f32)
    rs = np.array([np.sin(a30 + np.pi), np.sin(a30 - np.pi / 2), np.sin(a30), np.sin(a30 + np.pi / 2)], f32)
    alphas, ends = E.lidar_alphas(cfg, pos, obst, rc, rs)
    assert alphas.shape == (1, 1, 4)
    np.testing.assert_allclose(alphas[0, 0, 2], 0.2 / (0.5 * np.cos(a30)), rtol=1e-6)
    assert np.all(alphas[0, 0, [0, 1, 3]] == f32(1e6))                       # misses
    hits, idx = E.lidar_sense(cfg, pos, obst, rc, rs)
    assert idx[0, 0].tolist() == [2, 0]                                       # hit first, then stable order of misses
    np.testing.assert_allclose(hits[0, 0, 0], [0.7, 0.5 + 0.2 * np.tan(a30)], atol=1e-6)
    # a miss is NOT clamped: start + (end-start)*1e6  
    np.testing.assert_allclose(hits[0, 0, 1], [0.5 + 0.5e6 * rc[0], 0.5 + 0.5e6 * rs[0]], rtol=1e-5)


def test_parallel_ray_gives_nan_like_the_reference():
    """SURVEY A.3 corner / A.13 item 9: det == 0 exactly -> sign(det) = 0 -> x/0 -> 0*inf = NaN through min();
    NaN alphas sort last (argsort), so they only surface when fewer than k finite rays exist."""
    cfg = E.EnvCfg(E.LIDAR_SPREAD, n_agents=1, n_obs=1, n_rays=4, top_k=2)
    obst = _square(0.8, 0.5, 0.2)[None, None]
    pos = np.array([[[0.5, 0.5]]], f32)
    rc = np.array([-1, 0, 1, 0], f32)
    rs = np.array([0, -1, 0, 1], f32)
    alphas, _ = E.lidar_alphas(cfg, pos, obst, rc, rs)
    assert np.all(np.isnan(alphas))
    _, idx = E.lidar_sense(cfg, pos, obst, rc, rs)
    assert idx[0, 0].tolist() == [0, 1]


def test_start_inside_obstacle_zeroes_alpha():
    cfg = E.EnvCfg(E.LIDAR_SPREAD, n_agents=1, n_obs=1, n_rays=8, top_k=8)
    obst = _square(0.5, 0.5, 0.3, theta=0.3)[None, None]
    pos = np.array([[[0.52, 0.49]]], f32)
    rc, rs = E.ray_table(8)
    alphas, _ = E.lidar_alphas(cfg, pos, obst, rc, rs)
    assert np.all(alphas == 0)
    hits, idx = E.lidar_sense(cfg, pos, obst, rc, rs)
    assert idx[0, 0].tolist() == list(range(8))
    np.testing.assert_array_equal(hits[0, 0], np.broadcast_to(pos[0, 0], (8, 2)))


def test_all_miss_topk_is_stable():
    cfg = E.EnvCfg(E.LIDAR_SPREAD, n_agents=2, n_obs=1)
    obst = _square(1.4, 1.4, 0.1)[None, None]
    pos = np.array([[[0.2, 0.2], [0.3, 0.6]]], f32)
    hits, idx = E.lidar_sense(cfg, pos, obst, *E.ray_table(32))
    assert idx.shape == (1, 2, 8)
    assert np.all(idx == np.arange(8))


def test_rectangle_points_and_inside():
    rec = E.make_rect(np.array([1.0, 2.0], f32), f32(0.4), f32(0.2), f32(np.pi / 2))
    pts = rec[8:].reshape(4, 2)
    # rotated by 90 deg: (w/2,h/2)=(0.2,0.1) -> (-0.1, 0.2)
    np.testing.assert_allclose(pts[0], [1.0 - 0.1, 2.0 + 0.2], atol=1e-6)
    assert E.rect_inside(f32(1.0), f32(2.15), rec, 0.0)          # inside along the rotated long axis
    assert not E.rect_inside(f32(1.15), f32(2.0), rec, 0.0)
    assert E.rect_inside(f32(1.15), f32(2.0), rec, 0.06)         # inflated by r
    # rounded corner: just outside the corner circle
    assert not E.rect_inside(f32(1.0 + 0.1 + 0.05), f32(2.0 + 0.2 + 0.05), rec, 0.06)


def test_double_integrator_step_and_limits():
    cfg = E.EnvCfg(E.LIDAR_SPREAD, n_agents=2)
    agent = np.array([[[0.5, 0.5, 0.1, -0.2], [1.49, 0.01, 0.5, -0.5]]], f32)
    action = np.array([[[1.0, -1.0], [3.0, -3.0]]], f32)
    a = E.clip_action(action)
    assert a.max() == 1 and a.min() == -1
    nx = E.agent_step_euler(cfg, agent, a)
    np.testing.assert_allclose(nx[0, 0], [0.5 + 0.03 * 0.1, 0.5 - 0.03 * 0.2, 0.1 + 0.3, -0.2 - 0.3], rtol=1e-6)
    np.testing.assert_allclose(nx[0, 1], [1.5, 0.0, 0.5, -0.5], rtol=1e-6)     # clipped to area / vel limits
    mpe = E.EnvCfg(E.MPE_SPREAD, n_agents=2)
    nx = E.agent_step_euler(mpe, agent, a)
    np.testing.assert_allclose(nx[0, 1, 2:], [0.8, -0.8], rtol=1e-6)           # MPE velocity limit is +-1


def test_bicycle_step():
    cfg = E.EnvCfg(E.LIDAR_BICYCLE_TARGET, n_agents=1)
    th = 0.3
    agent = np.array([[[0.5, 0.5, np.cos(th), np.sin(th), 0.4]]], f32)
    a = np.array([[[0.5, -1.0]]], f32)
    nx = E.agent_step_euler(cfg, agent, a)[0, 0]
    thn = th + 0.4 * 0.5 * 0.03 * 10
    np.testing.assert_allclose(nx, [0.5 + 0.4 * np.cos(th) * 0.03, 0.5 + 0.4 * np.sin(th) * 0.03,
                                    np.cos(thn), np.sin(thn), 0.4 - 0.3], rtol=1e-5)
    np.testing.assert_allclose(E.state2feat(cfg, agent)[0, 0], [0.5, 0.5, 0.4 * np.cos(th), 0.4 * np.sin(th)], rtol=1e-6)


def test_cost_two_agents_closed_form():
    cfg = E.EnvCfg(E.LIDAR_SPREAD, n_agents=2, n_obs=1, top_k=2)
    agent = np.array([[[0.5, 0.5, 0, 0], [0.53, 0.54, 0, 0]]], f32)      # distance 0.05 < 2r: colliding
    hits = np.array([[[[0.5, 0.53], [9, 9]], [[0.53, 1.0], [0.53, 0.9]]]], f32)
    c = E.get_cost(cfg, agent, hits)
    np.testing.assert_allclose(c[0, :, 0], [0.1 - 0.05 + 0.5] * 2, atol=1e-6)
    np.testing.assert_allclose(c[0, 0, 1], (0.05 - 0.03) + 0.5, atol=1e-6)   # unsafe: positive -> +0.5
    np.testing.assert_allclose(c[0, 1, 1], (0.05 - 0.36) - 0.5, atol=1e-6)
    # LiDAR clips to [-1, 1]; MPE only from below
    far = np.array([[[0.1, 0.1, 0, 0], [1.4, 1.4, 0, 0]]], f32)
    assert E.get_cost(cfg, far, hits)[0, 0, 0] == -1.0
    mpe = E.EnvCfg(E.MPE_SPREAD, n_agents=2, n_obs=1)
    obs = np.array([[[0.1, 0.1, 0, 0]]], f32)
    cm = E.get_cost(mpe, far, obs)
    np.testing.assert_allclose(cm[0, 0, 1], 0.1 + 0.5, atol=1e-6)


def test_reward_spread_vs_target():
    agent = np.array([[[0.0, 0.0, 0, 0], [1.0, 0.0, 0, 0]]], f32)
    goal = np.array([[[1.0, 0.1, 0, 0], [0.0, 0.0, 0, 0]]], f32)
    action = np.array([[[1.0, 0.0], [0.0, 0.5]]], f32)
    sp = E.EnvCfg(E.LIDAR_SPREAD, n_agents=2)
    tg = E.EnvCfg(E.LIDAR_TARGET, n_agents=2)
    r_sp = E.get_reward(sp, agent, goal, action)[0]
    r_tg = E.get_reward(tg, agent, goal, action)[0]
    act_pen = (1.0 + 0.25) / 2 * 1e-4
    np.testing.assert_allclose(r_sp, -(0.1 + 0.0) / 2 * 0.01 - 0.5 * 0.001 - act_pen, rtol=1e-5)
    d_t = (np.hypot(1.0, 0.1) + 1.0) / 2
    np.testing.assert_allclose(r_tg, -d_t * 0.01 - 1.0 * 0.001 - act_pen, rtol=1e-5)


@pytest.mark.parametrize("kind,n,n_obs", [(E.LIDAR_SPREAD, 8, 3), (E.LIDAR_TARGET, 4, 2), (E.LIDAR_BICYCLE_TARGET, 16, 8),
                                           (E.MPE_SPREAD, 3, 3), (E.MPE_TARGET, 3, 0), (E.MPE_TARGET, 3, 3),
                                           (E.LIDAR_SPREAD, 2, 0)])
def test_graph_layout_contract(kind, n, n_obs):
    """shape/layout contracts of SURVEY §8 table and A.2."""
    cfg = E.EnvCfg(kind, n_agents=n, n_obs=n_obs)
    B = 3
    agent, goal, obst = E.env_reset(cfg, [1, 2, 3])
    hits = None
    if cfg.is_lidar and n_obs > 0:
        hits, _ = E.lidar_sense(cfg, agent[..., :2], obst, *E.ray_table(cfg.n_rays))
    g = E.get_graph(cfg, agent, goal, obst, hits)
    N, Eg = cfg.num_nodes, cfg.num_edges
    expected = {(E.LIDAR_SPREAD, 8, 3): (81, 192), (E.LIDAR_BICYCLE_TARGET, 16, 8): (161, 400), (E.MPE_SPREAD, 3, 3): (10, 27),
                (E.MPE_TARGET, 3, 0): (7, 12)}
    if (kind, n, n_obs) in expected:
        assert (N, Eg) == expected[(kind, n, n_obs)]
    assert g["nodes"].shape == (B, N, cfg.node_dim) and g["edges"].shape == (B, Eg, 4)
    assert g["states"].shape == (B, N, cfg.state_dim)
    assert g["receivers"].shape == g["senders"].shape == (B, Eg)
    pad = N - 1
    assert np.all(g["node_type"][:, pad] == -1) and np.all(g["states"][:, pad] == -1) and np.all(g["nodes"][:, pad] == 0)
    assert np.all(g["node_type"][:, :n] == 0) and np.all(g["node_type"][:, n:2 * n] == 1)
    sd = cfg.state_dim
    assert np.all(g["nodes"][:, :n, sd + 2] == 1) and np.all(g["nodes"][:, n:2 * n, sd + 1] == 1)
    if cfg.n_obs_nodes:
        assert np.all(g["nodes"][:, 2 * n:pad, sd] == 1) and np.all(g["node_type"][:, 2 * n:pad] == 2)
    r, s = g["receivers"], g["senders"]
    assert r.min() >= 0 and r.max() <= pad and s.min() >= 0 and s.max() <= pad
    assert np.all((r == pad) == (s == pad))                     # masked => both endpoints re-routed to the pad node
    assert np.all(r[r != pad] < n)                              # only agents receive
    aa = slice(0, n * n)
    diag = np.arange(n) * (n + 1)
    assert np.all(r[:, aa][:, diag] == pad)                     # self edges always masked
    ge = slice(n * n, n * n + cfg.n_goal_edges)
    assert np.all(r[:, ge] != pad)                              # agent-goal edges never masked
    # edge feature = feat(receiver) - feat(sender) even when masked (graph.py:41)
    fa = E.state2feat(cfg, agent)
    np.testing.assert_array_equal(g["edges"][:, 1], fa[:, 0] - fa[:, 1])


def test_reset_respects_separation():
    cfg = E.EnvCfg(E.LIDAR_SPREAD, n_agents=8, n_obs=3)
    agent, goal, obst = E.env_reset(cfg, [11, 12])
    for b in range(2):
        for arr in (agent, goal):
            p = arr[b, :, :2]
            d = np.linalg.norm(p[:, None] - p[None], axis=-1) + np.eye(8) * 10
            assert d.min() > 2.2 * 0.05
            assert np.linalg.norm(p, axis=-1).min() > 0.11           # origin neighbourhood excluded
            for o in range(3):
                assert not np.any(E.rect_inside(p[:, 0], p[:, 1], obst[b, o], 0.055))
        assert np.all(agent[b, :, 2:] == 0)
    w = obst[..., 2:4]
    assert w.min() >= 0.1 and w.max() <= 0.3


def test_step_uses_pre_step_graph_for_reward_and_cost():
    cfg = E.EnvCfg(E.LIDAR_SPREAD, n_agents=3, n_obs=2)
    agent, goal, obst = E.env_reset(cfg, [5])
    tab = E.ray_table(32)
    hits, _ = E.lidar_sense(cfg, agent[..., :2], obst, *tab)
    act = np.ones((1, 3, 2), f32) * 0.5
    out = E.env_step(cfg, agent, goal, obst, hits, act, tab)
    np.testing.assert_array_equal(out["cost"], E.get_cost(cfg, agent, hits))
    np.testing.assert_array_equal(out["reward"], E.get_reward(cfg, agent, goal, act))
    h2, _ = E.lidar_sense(cfg, out["next_agent"][..., :2], obst, *tab)
    np.testing.assert_array_equal(out["next_hits"], h2)
