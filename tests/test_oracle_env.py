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


# ---- task variants: closed-form known answers ---------------------------------------------------------
def test_line_goals_divide_the_segment():
    """landmark2goal: lidar_line.py:131-136 (ends included), mpe_line.py:124-133 (n <= 3: interior points)."""
    lm = np.array([[[0.0, 0.0, 0, 0], [1.0, 0.5, 0, 0]]], f32)
    g5 = E.reward_goal_positions(E.EnvCfg(E.MPE_LINE, n_agents=5), lm)[0]
    np.testing.assert_allclose(g5, [[0, 0], [.25, .125], [.5, .25], [.75, .375], [1, .5]], atol=1e-7)
    g3 = E.reward_goal_positions(E.EnvCfg(E.MPE_LINE, n_agents=3), lm)[0]
    np.testing.assert_allclose(g3, [[.25, .125], [.5, .25], [.75, .375]], atol=1e-7)
    gl = E.reward_goal_positions(E.EnvCfg(E.LIDAR_LINE, n_agents=3), lm)[0]          # LidarLine: always the end-point form
    np.testing.assert_allclose(gl, [[0, 0], [.5, .25], [1, .5]], atol=1e-7)
    assert E.EnvCfg(E.MPE_LINE, n_agents=5).n_goals == 2 == E.EnvCfg(E.LIDAR_LINE, n_agents=5).n_goals


def test_formation_goals_lie_on_the_comm_circle():
    """mpe_formation.py:94-98: landmark + R [cos, sin](2 pi i / n)."""
    cfg = E.EnvCfg(E.MPE_FORMATION, n_agents=4)
    g = E.reward_goal_positions(cfg, np.array([[[1.0, 1.0, 0, 0]]], f32))[0]
    np.testing.assert_allclose(g, [[1.5, 1.0], [1.0, 1.5], [0.5, 1.0], [1.0, 0.5]], atol=1e-6)
    assert cfg.n_goals == 1 and cfg.num_nodes == 4 + 1 + 3 + 1


def test_line_reward_closed_form():
    """agents sitting exactly on the 3 interior goals except one displaced by 0.1: reward = -(0.1/3) 0.01 - (1/3) 0.001 - action."""
    cfg = E.EnvCfg(E.MPE_LINE, n_agents=3)
    lm = np.array([[[0.0, 0.0, 0, 0], [1.0, 0.0, 0, 0]]], f32)
    agent = np.array([[[0.25, 0, 0, 0], [0.5, 0.1, 0, 0], [0.75, 0, 0, 0]]], f32)
    act = np.zeros((1, 3, 2), f32); act[0, 0] = [0.6, 0.8]
    r = float(E.get_reward(cfg, agent, lm, act)[0])
    assert abs(r - (-(0.1 / 3) * 0.01 - (1 / 3) * 0.001 - (1.0 / 3) * 0.0001)) < 1e-8


def test_connect_spread_third_cost_and_two_sided_clip():
    """mpe_connect_spread.py:103-136: connectivity = max_i(nearest-neighbour distance_i) - connect_radius for every agent; all three
    components get the +-0.5 margin and are clipped to [-1, 1] (the MPE base clips from below only)."""
    cfg = E.EnvCfg(E.MPE_CONNECT_SPREAD, n_agents=3)
    assert (cfg.n_obs, cfg.n_cost, cfg.area_size) == (1, 3, 1.0) and abs(cfg.obs_radius - 0.25) < 1e-9
    agent = np.array([[[0.1, 0.1, 0, 0], [0.3, 0.1, 0, 0], [0.3, 0.8, 0, 0]]], f32)     # nn distances 0.2, 0.2, 0.7
    obs = np.array([[[0.9, 0.5, 0, 0]]], f32)
    c = E.get_cost(cfg, agent, obs)[0]
    assert c.shape == (3, 3)
    np.testing.assert_allclose(c[:, 2], np.clip((0.7 - 0.45) + 0.5, -1, 1), atol=1e-6)   # disconnected: positive for all
    np.testing.assert_allclose(c[:, 0], np.clip(np.array([0.1 - 0.2, 0.1 - 0.2, 0.1 - 0.7]) - 0.5, -1, 1), atol=1e-6)
    close = agent.copy(); close[0, 2, :2] = [0.3, 0.3]                                  # all within 0.45: connected
    c2 = E.get_cost(cfg, close, obs)[0]
    np.testing.assert_allclose(c2[:, 2], (0.2 - 0.45) - 0.5, atol=1e-6)
    touching = agent.copy(); touching[0, 1, :2] = [0.1, 0.1]                            # collision: 0.1 + 0.5 stays 0.6; overlap on an
    obs_hit = np.array([[[0.1, 0.1, 0, 0]]], f32)                                        # obstacle: 0.3 + 0.5 = 0.8 (no clip needed), but
    assert float(E.get_cost(cfg, touching, obs_hit)[0, 0, 1]) == pytest.approx(0.8, abs=1e-6)
    far = E.EnvCfg(E.MPE_SPREAD, n_agents=3, n_obs=1)                                     # the base class clips from below only:
    assert float(E.get_cost(far, touching, obs_hit)[0, 0, 1]) == pytest.approx(0.6, abs=1e-6)   # car + obs radius 0.1 + 0.5


def test_corridor_layout_masks_and_limits():
    """mpe_corridor.py:35-39,55-56,62-65,93: obs_radius = (A - width) / 4, discs at (r, A/2) and (A - r, A/2), y may reach 2 A,
    agent-obstacle edges are never masked."""
    cfg = E.EnvCfg(E.MPE_CORRIDOR, n_agents=3, n_obs=7)
    assert cfg.n_obs == 2 and abs(cfg.obs_radius - 0.2) < 1e-9 and cfg.area_size == 1.0
    a, g, o = E.env_reset(cfg, np.array([5, 6], dtype=np.int64))
    np.testing.assert_allclose(o[0, :, :2], [[0.2, 0.5], [0.8, 0.5]], atol=1e-7)
    th = E.reset_thresholds(cfg)
    assert (a[..., 1] <= th["side_y"]).all() and (g[..., 1] >= th["goal_shift_y"]).all()  # agents below, goals above the walls
    lo, hi = E.state_limits(cfg)
    assert hi[1] == f32(2.0) and hi[0] == f32(1.0)
    far = a.copy(); far[..., :2] = [[0.0, 0.0], [0.0, 1.9], [1.0, 1.9]]
    gr = E.get_graph(cfg, far, g, o, None)
    n = 3
    assert (gr["senders"][:, n * n + n * n:] != cfg.num_nodes - 1).all()


@pytest.mark.parametrize("kind,n,n_obs", [(E.LIDAR_LINE, 4, 2), (E.MPE_LINE, 3, 3), (E.MPE_LINE, 5, 3), (E.MPE_FORMATION, 4, 3),
                                           (E.MPE_CONNECT_SPREAD, 4, 1)])
def test_variant_reset_invariants(kind, n, n_obs):
    cfg = E.EnvCfg(kind, n_agents=n, n_obs=n_obs)
    th = E.reset_thresholds(cfg)
    a, g, o = E.env_reset(cfg, np.arange(11, 19, dtype=np.int64))
    assert a.shape == (8, n, 4) and g.shape == (8, cfg.n_goals, 4) and (a[..., 2:] == 0).all() and (g[..., 2:] == 0).all()
    for b in range(8):
        d = np.linalg.norm(a[b, :, None, :2] - a[b, None, :, :2], axis=-1) + np.eye(n) * 9
        assert d.min() > th["min_dist"]
        rg = E.reward_goal_positions(cfg, g[b:b + 1])[0]
        if kind in (E.LIDAR_LINE, E.MPE_LINE):
            assert np.linalg.norm(g[b, 1, :2] - g[b, 0, :2]) >= th["line_min_dist"]
        if kind == E.MPE_FORMATION:
            lo = cfg.comm_radius + 2 * cfg.car_radius
            assert (g[b, 0, :2] >= lo - 1e-6).all() and (g[b, 0, :2] <= cfg.area_size - lo + 1e-6).all()
        if kind == E.MPE_CONNECT_SPREAD:
            assert (np.sort(d, 1)[:, 0] <= cfg.connect_radius).all()
            gd = np.linalg.norm(g[b, :, None, :2] - g[b, None, :, :2], axis=-1) + np.eye(n) * 9
            assert (gd.min(1) <= cfg.connect_radius).all()
            assert o[b, 0, 1] == f32(0.5) and cfg.obs_radius <= o[b, 0, 0] <= 1 - cfg.obs_radius
        elif kind == E.LIDAR_LINE:
            pts = np.concatenate([a[b, :, :2], rg], 0)
            assert not E.rect_inside(pts[:, None, 0], pts[:, None, 1], o[b][None], f32(0.05 * 1.1)).any()
        else:
            do = np.linalg.norm(o[b, :, None, :2] - a[b, None, :, :2], axis=-1)
            dg = np.linalg.norm(o[b, :, None, :2] - rg[None], axis=-1)
            assert (do > cfg.car_radius + cfg.obs_radius).all() and (dg > 2 * cfg.car_radius + cfg.obs_radius).all()
