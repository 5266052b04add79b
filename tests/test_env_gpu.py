"""GPU parity: dgppo_env_step / dgppo_env_reset / dgppo_graph_materialize (HIP, through the C ABI) vs oracle/env_np.py.
Double-integrator and MPE families are bit-exact (kernels built with -ffp-contract=off); the bicycle's dynamics use
device atan2/sin/cos and are compared within 1e-6, after which sensing/graph on identical states is bit-exact again."""
import numpy as np
import pytest
import torch

from oracle import env_np as E

pytestmark = pytest.mark.gpu
f32 = np.float32

CASES = [
    ("LidarSpread", 8, 3), ("LidarSpread", 3, 1), ("LidarTarget", 4, 2), ("LidarSpread", 2, 0),
    ("MPESpread", 3, 3), ("MPETarget", 3, 0), ("MPETarget", 3, 3), ("MPESpread", 5, 2),
    # the remaining instantiations of the wave-per-env kernel (csrc/env_wave.hip)
    ("LidarTarget", 8, 3), ("LidarSpread", 4, 2), ("LidarSpread", 16, 8),
]


def _mk(kind_name, n, n_obs):
    from dgppo_amd import _native as N
    kind = N.ENV_KINDS[kind_name]
    return N.make_env_cfg(kind, n, n_obs), E.EnvCfg(kind, n_agents=n, n_obs=n_obs)


def _random_state(ocfg, B, seed):
    """random but valid scene: oracle reset for the scene, then random velocities / displaced agents so masks vary."""
    rng = np.random.default_rng(seed)
    agent, goal, obst = E.env_reset(ocfg, rng.integers(1, 2 ** 62, size=B))
    agent = agent.copy()
    # pull agents towards each other / obstacles for some envs so that radius masks and LiDAR hits trigger
    agent[:, :, :2] = (agent[:, :, :2] * 0.5 + 0.4).astype(f32)
    v = ocfg.vel_limit
    if ocfg.is_bicycle:
        th = rng.uniform(0, 2 * np.pi, size=agent.shape[:2])
        agent[..., 2] = np.cos(th)
        agent[..., 3] = np.sin(th)
        agent[..., 4] = rng.uniform(-0.5, 0.5, size=agent.shape[:2])
    else:
        agent[..., 2:4] = rng.uniform(-v, v, size=agent.shape[:2] + (2,))
    action = rng.uniform(-1.5, 1.5, size=agent.shape[:2] + (2,)).astype(f32)
    return agent.astype(f32), goal, obst, action


def _to(x, dev):
    return None if x is None else torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def _run_step(cfg, ocfg, agent, goal, obst, hits, action, dev, want_graph=True):
    from dgppo_amd import ops_env as O
    B, n = agent.shape[:2]
    rc, rs = O.ray_tables(max(cfg.n_rays, 1), dev) if cfg.is_lidar else (None, None)
    nx = torch.empty(B, n, cfg.state_dim, device=dev)
    nh = torch.empty(B, n, cfg.top_k, 2, device=dev) if (cfg.is_lidar and cfg.n_obs > 0) else None
    rew = torch.empty(B, device=dev)
    cost = torch.empty(B, n, cfg.n_cost, device=dev)
    g = O.alloc_graph(cfg, B, dev) if want_graph else None
    O.env_step(cfg, _to(agent, dev), _to(action, dev), _to(goal, dev), _to(obst, dev), _to(hits, dev), rc, rs,
               nx, nh, rew if action is not None else None, cost if action is not None else None, g)
    torch.cuda.synchronize()
    out = dict(next_agent=nx.cpu().numpy(), next_hits=None if nh is None else nh.cpu().numpy(),
               reward=rew.cpu().numpy(), cost=cost.cpu().numpy())
    if g is not None:
        out["graph"] = {k: v.cpu().numpy() for k, v in g.items()}
    return out


def _assert_graph_equal(got, want):
    for k in ("receivers", "senders", "node_type", "n_node", "n_edge"):
        np.testing.assert_array_equal(got[k], want[k], err_msg=k)          # integer outputs: bit-exact
    for k in ("nodes", "edges", "states"):
        np.testing.assert_array_equal(got[k].view(np.uint32), want[k].view(np.uint32), err_msg=k)


@pytest.mark.parametrize("kind,n,n_obs", CASES)
def test_step_bit_exact(cuda, kind, n, n_obs):
    cfg, ocfg = _mk(kind, n, n_obs)
    B = 64
    agent, goal, obst, action = _random_state(ocfg, B, seed=hash((kind, n, n_obs)) % 1000)
    tab = E.ray_table(ocfg.n_rays)
    hits = None
    if ocfg.is_lidar and n_obs > 0:
        hits, _ = E.lidar_sense(ocfg, agent[..., :2], obst, *tab)
    want = E.env_step(ocfg, agent, goal, obst, hits, action, tab)
    got = _run_step(cfg, ocfg, agent, goal, obst, hits, action, cuda)
    np.testing.assert_array_equal(got["next_agent"].view(np.uint32), want["next_agent"].view(np.uint32))
    np.testing.assert_array_equal(got["reward"].view(np.uint32), want["reward"].view(np.uint32))
    np.testing.assert_array_equal(got["cost"].view(np.uint32), want["cost"].view(np.uint32))
    if want["next_hits"] is not None:
        np.testing.assert_array_equal(got["next_hits"].view(np.uint32), want["next_hits"].view(np.uint32))
    _assert_graph_equal(got["graph"], want["graph"])
    # masks must actually vary in this test, otherwise it proves little
    r = want["graph"]["receivers"]
    pad = ocfg.num_nodes - 1
    assert (r == pad).any() and (r[:, :n * n] != pad).any()


def test_bicycle_dynamics_tolerance_then_exact_sensing(cuda):
    cfg, ocfg = _mk("LidarBicycleTarget", 16, 8)
    B = 32
    agent, goal, obst, action = _random_state(ocfg, B, seed=7)
    tab = E.ray_table(32)
    hits, _ = E.lidar_sense(ocfg, agent[..., :2], obst, *tab)
    want = E.env_step(ocfg, agent, goal, obst, hits, action, tab)
    got = _run_step(cfg, ocfg, agent, goal, obst, hits, action, cuda)
    np.testing.assert_allclose(got["next_agent"], want["next_agent"], atol=1e-6, rtol=0)   # device atan2/sincos
    np.testing.assert_array_equal(got["cost"], want["cost"])
    np.testing.assert_array_equal(got["reward"], want["reward"])
    # sense-only on the DEVICE's next state: everything downstream is bit-exact again
    nx = got["next_agent"]
    h2, _ = E.lidar_sense(ocfg, nx[..., :2], obst, *tab)
    g2 = E.get_graph(ocfg, nx, goal, obst, h2)
    np.testing.assert_array_equal(got["next_hits"].view(np.uint32), h2.view(np.uint32))
    _assert_graph_equal(got["graph"], g2)
    assert got["graph"]["nodes"].shape == (B, 161, 8) and got["graph"]["edges"].shape == (B, 400, 4)


def test_sense_only_and_materialize(cuda):
    from dgppo_amd import ops_env as O
    cfg, ocfg = _mk("LidarSpread", 8, 3)
    B = 16
    agent, goal, obst, _ = _random_state(ocfg, B, seed=3)
    tab = E.ray_table(32)
    got = _run_step(cfg, ocfg, agent, goal, obst, None, None, cuda)
    h, _ = E.lidar_sense(ocfg, agent[..., :2], obst, *tab)
    np.testing.assert_array_equal(got["next_agent"], agent)
    np.testing.assert_array_equal(got["next_hits"].view(np.uint32), h.view(np.uint32))
    _assert_graph_equal(got["graph"], E.get_graph(ocfg, agent, goal, obst, h))
    g = O.alloc_graph(cfg, B, cuda)
    O.graph_materialize(cfg, _to(agent, cuda), _to(goal, cuda), _to(obst, cuda), _to(h, cuda), g)
    torch.cuda.synchronize()
    _assert_graph_equal({k: v.cpu().numpy() for k, v in g.items()}, E.get_graph(ocfg, agent, goal, obst, h))


@pytest.mark.parametrize("kind,n,n_obs", [("LidarSpread", 8, 3), ("LidarBicycleTarget", 4, 3), ("MPESpread", 3, 3),
                                           ("MPETarget", 3, 0)])
def test_reset_matches_oracle_stream(cuda, kind, n, n_obs):
    from dgppo_amd import ops_env as O
    cfg, ocfg = _mk(kind, n, n_obs)
    B = 48
    seeds = (np.arange(B, dtype=np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
    seeds_t = torch.from_numpy(seeds.view(np.int64)).to(cuda)
    agent = torch.empty(B, n, cfg.state_dim, device=cuda)
    goal = torch.empty(B, n, cfg.state_dim, device=cuda)
    obst = torch.empty(B, n_obs, cfg.obst_stride, device=cuda) if n_obs > 0 else None
    O.env_reset(cfg, seeds_t, agent, goal, obst)
    torch.cuda.synchronize()
    wa, wg, wo = E.env_reset(ocfg, [int(s) for s in seeds])
    bad = np.nonzero((agent.cpu().numpy()[..., :2] != wa[..., :2]).any(axis=(1, 2)))[0]
    if len(bad):
        print("mismatching envs", bad.tolist())
        for b in bad[:3]:
            print("env", b, "gpu agent", agent.cpu().numpy()[b, :, :2].tolist(), "oracle", wa[b, :, :2].tolist(),
                  "gpu goal", goal.cpu().numpy()[b, :, :2].tolist(), "oracle", wg[b, :, :2].tolist())
    if kind == "LidarBicycleTarget":
        np.testing.assert_array_equal(agent.cpu().numpy()[..., :2], wa[..., :2])
        np.testing.assert_allclose(agent.cpu().numpy(), wa, atol=1e-6)
    else:
        np.testing.assert_array_equal(agent.cpu().numpy(), wa)          # integer RNG stream + exact fp32 arithmetic
    np.testing.assert_array_equal(goal.cpu().numpy(), wg)
    if n_obs > 0:
        go = obst.cpu().numpy()
        if ocfg.is_lidar:
            np.testing.assert_array_equal(go[..., :5], wo[..., :5])      # centre, w, h, theta from the integer stream
            np.testing.assert_allclose(go, wo, atol=1e-6)                # cos/sin/points: device trig
        else:
            np.testing.assert_array_equal(go, wo)


def test_reset_reports_scenes_it_could_not_place(cuda):
    """The reference's rejection loops are unbounded; the kernels bound them.  (1) A density at which placement cannot succeed is
    refused on the host before anything is launched.  (2) A config that passes that test but still cannot be placed — here
    obstacles cover the whole area — finishes (every loop is bounded) and says so: the failure counter equals the number
    of invalid scenes, and the API-level reset raises instead of handing out a colliding scene.  (3) The benchmark
    configs leave the counter at zero."""
    from dgppo_amd import _native as N, ops_env as O
    crowded = N.make_env_cfg(N.ENV_KINDS["MPESpread"], 16, 0, area_size=0.3)
    seeds = torch.arange(1, 9, dtype=torch.int64, device=cuda)
    with pytest.raises(ValueError, match="cannot be placed"):
        O.env_reset(crowded, seeds, torch.empty(8, 16, 4, device=cuda), torch.empty(8, 16, 4, device=cuda), None)
    walled = N.make_env_cfg(N.ENV_KINDS["LidarSpread"], 2, 48, area_size=0.4)
    nf = torch.zeros(1, dtype=torch.int32, device=cuda)
    O.env_reset(walled, seeds, torch.empty(8, 2, 4, device=cuda), torch.empty(8, 2, 4, device=cuda),
                torch.empty(8, 48, 16, device=cuda), nf)
    assert int(nf.item()) == 8
    cfg, _ = _mk("LidarSpread", 8, 3)
    nf.zero_()
    big = torch.arange(1, 4097, dtype=torch.int64, device=cuda) * 7919
    O.env_reset(cfg, big, torch.empty(4096, 8, 4, device=cuda), torch.empty(4096, 8, 4, device=cuda),
                torch.empty(4096, 3, 16, device=cuda), nf)
    assert int(nf.item()) == 0


def test_full_size_rollout_properties(cuda):
    """BASELINE config 3 at full size (4096 envs): 3 chained steps on the device vs the oracle, every output, bit-exact."""
    from dgppo_amd import ops_env as O
    cfg, ocfg = _mk("LidarSpread", 8, 3)
    B, n = 4096, 8
    rng = np.random.default_rng(0)
    seeds = rng.integers(1, 2 ** 62, size=B).astype(np.int64)
    agent = torch.empty(B, n, 4, device=cuda)
    goal = torch.empty(B, n, 4, device=cuda)
    obst = torch.empty(B, 3, 16, device=cuda)
    O.env_reset(cfg, torch.from_numpy(seeds).to(cuda), agent, goal, obst)
    rc, rs = O.ray_tables(32, cuda)
    hits = torch.empty(B, n, 8, 2, device=cuda)
    O.env_step(cfg, agent, None, goal, obst, None, rc, rs, None, hits, None, None, None)
    a_np, g_np, o_np, h_np = (x.cpu().numpy() for x in (agent, goal, obst, hits))
    tab = E.ray_table(32)
    for t in range(3):
        action = rng.uniform(-1.2, 1.2, size=(B, n, 2)).astype(f32)
        got = _run_step(cfg, ocfg, a_np, g_np, o_np, h_np, action, cuda)
        want = E.env_step(ocfg, a_np, g_np, o_np, h_np, action, tab)
        for k in ("next_agent", "next_hits", "reward", "cost"):
            np.testing.assert_array_equal(got[k].view(np.uint32), want[k].view(np.uint32), err_msg=f"{k} t={t}")
        _assert_graph_equal(got["graph"], want["graph"])
        a_np, h_np = got["next_agent"], got["next_hits"]
    # size-independent properties
    g = got["graph"]
    assert np.all(g["n_node"] == 81) and np.all(g["n_edge"] == 192)
    assert np.all((g["receivers"] == 80) == (g["senders"] == 80))
    assert np.all(g["states"][:, 80] == -1)


def test_randn_rows_is_a_column_window_of_the_flat_stream(cuda):
    """dgppo_randn_rows: out[r, c] = element r * global_row_len + col_offset + c of dgppo_randn(seed, 0) — every alignment of
    the window against the 4-wide Philox blocks, the full row (== the flat stream), and the union of two ranks' windows."""
    from dgppo_amd import ops_env as O
    rows, L = 5, 38                                    # 38 % 4 != 0: rows start at every phase of a Philox block
    flat = torch.empty(rows * L, device=cuda)
    O.randn(99, 0, flat)
    full = torch.empty(rows, L, device=cuda)
    O.randn_rows(99, full, L, 0)
    assert torch.equal(full.view(-1), flat)
    for off, ln in ((0, 19), (19, 19), (1, 4), (3, 1), (7, 30), (37, 1), (6, 0)):
        w = torch.full((rows, ln), float("nan"), device=cuda)
        O.randn_rows(99, w, L, off)
        assert torch.equal(w, flat.view(rows, L)[:, off:off + ln]), (off, ln)
    with pytest.raises(ValueError):
        O.randn_rows(99, torch.empty(rows, 10, device=cuda), L, 30)     # window runs past the row


def test_randn_moments_and_determinism(cuda):
    from dgppo_amd import ops_env as O
    x = torch.empty(1 << 20, device=cuda)
    y = torch.empty(1 << 20, device=cuda)
    O.randn(123, 0, x)
    O.randn(123, 0, y)
    torch.cuda.synchronize()
    assert torch.equal(x, y)
    assert abs(x.mean().item()) < 5e-3 and abs(x.std().item() - 1.0) < 5e-3
    assert torch.isfinite(x).all()
    O.randn(124, 0, y)
    assert not torch.equal(x, y)


def test_generic_kernel_path_matches_specialised(cuda, monkeypatch):
    """LiDAR envs normally run the specialised n_rays == 32 kernel; the generic kernel (other ray counts, forced here with
    DGPPO_GENERIC_ENV_KERNEL) must give the same bits, and so must a non-32 ray fan against the oracle."""
    cfg, ocfg = _mk("LidarSpread", 8, 3)
    B = 96
    agent, goal, obst, action = _random_state(ocfg, B, seed=21)
    tab = E.ray_table(32)
    hits, _ = E.lidar_sense(ocfg, agent[..., :2], obst, *tab)
    fast = _run_step(cfg, ocfg, agent, goal, obst, hits, action, cuda)
    monkeypatch.setenv("DGPPO_GENERIC_ENV_KERNEL", "1")
    gen = _run_step(cfg, ocfg, agent, goal, obst, hits, action, cuda)
    monkeypatch.delenv("DGPPO_GENERIC_ENV_KERNEL")
    for k in ("next_agent", "next_hits", "reward", "cost"):
        np.testing.assert_array_equal(fast[k].view(np.uint32), gen[k].view(np.uint32), err_msg=k)
    _assert_graph_equal(fast["graph"], gen["graph"])
    # 16 rays, top-4: generic kernel vs oracle
    from dgppo_amd import _native as N, ops_env as O
    cfg16 = N.make_env_cfg(0, 5, 2, n_rays=16, top_k=4)
    o16 = E.EnvCfg(0, n_agents=5, n_obs=2, n_rays=16, top_k=4)
    agent, goal, obst, action = _random_state(o16, 32, seed=22)
    tab16 = E.ray_table(16)
    hits, _ = E.lidar_sense(o16, agent[..., :2], obst, *tab16)
    want = E.env_step(o16, agent, goal, obst, hits, action, tab16)
    rc, rs = O.ray_tables(16, cuda)
    nx = torch.empty(32, 5, 4, device=cuda); nh = torch.empty(32, 5, 4, 2, device=cuda)
    rew = torch.empty(32, device=cuda); cost = torch.empty(32, 5, 2, device=cuda)
    g = O.alloc_graph(cfg16, 32, cuda)
    O.env_step(cfg16, _to(agent, cuda), _to(action, cuda), _to(goal, cuda), _to(obst, cuda), _to(hits, cuda), rc, rs, nx, nh, rew, cost, g)
    np.testing.assert_array_equal(nh.cpu().numpy().view(np.uint32), want["next_hits"].view(np.uint32))
    np.testing.assert_array_equal(cost.cpu().numpy().view(np.uint32), want["cost"].view(np.uint32))
    _assert_graph_equal({k: v.cpu().numpy() for k, v in g.items()}, want["graph"])


def _adversarial_state(ocfg, B, seed):
    """_random_state plus the corner cases of the ray-cast / top-k: agents INSIDE an obstacle (all 32 alphas are 0: ranks
    are the ray indices), agents far from everything (every obstacle culled), two agents on the same spot, agents on the
    area boundary, and a NaN hit point in the pre-step graph (the obstacle cost must come out NaN, like jnp.min)."""
    agent, goal, obst, action = _random_state(ocfg, B, seed)
    n = ocfg.n_agents
    c = obst[:, :, 0:2]
    for e in range(0, B, 7):
        agent[e, e % n, :2] = c[e, e % ocfg.n_obs]                     # inside obstacle (its centre)
        agent[e, e % n, 2:4] = 0.0
    for e in range(1, B, 7):
        agent[e, :, :2] = np.float32(1.49)                             # all agents piled in a corner, far from most obstacles
    for e in range(2, B, 7):
        agent[e, 1, :2] = agent[e, 0, :2]                              # coincident agents (distance exactly 0)
    for e in range(3, B, 7):
        agent[e, 0, :2] = 0.0                                          # on the boundary
    return agent, goal, obst, action


@pytest.mark.parametrize("kind,n,n_obs,B", [("LidarSpread", 8, 3, 1500), ("LidarTarget", 8, 3, 300),
                                            ("LidarBicycleTarget", 16, 8, 200), ("LidarSpread", 4, 2, 333)])
def test_wave_kernel_equals_workgroup_kernel_and_oracle(cuda, monkeypatch, kind, n, n_obs, B):
    """csrc/env_wave.hip (wave-per-env, obstacle culling, ballot top-k, squared-distance minima and masks) against the
    workgroup-per-env kernel of env_step.hip (forced with DGPPO_NO_WAVE_ENV_KERNEL) — every output, every bit, in all three
    modes (step / sense-only / materialise), with persistent waves looping over several envs — and against the oracle."""
    cfg, ocfg = _mk(kind, n, n_obs)
    agent, goal, obst, action = _adversarial_state(ocfg, B, seed=5)
    tab = E.ray_table(32)
    hits, _ = E.lidar_sense(ocfg, agent[..., :2], obst, *tab)
    hits = hits.copy()
    hits[5, 0, 3, 0] = np.nan                                           # a NaN hit point in graph_t
    for wave_envs in ("1", "3"):
        monkeypatch.setenv("DGPPO_WAVE_ENVS", wave_envs)                # 3: every wave walks >= 3 environments
        fast = _run_step(cfg, ocfg, agent, goal, obst, hits, action, cuda)
        fast_sense = _run_step(cfg, ocfg, agent, goal, obst, None, None, cuda)
        monkeypatch.setenv("DGPPO_NO_WAVE_ENV_KERNEL", "1")
        ref = _run_step(cfg, ocfg, agent, goal, obst, hits, action, cuda)
        ref_sense = _run_step(cfg, ocfg, agent, goal, obst, None, None, cuda)
        monkeypatch.delenv("DGPPO_NO_WAVE_ENV_KERNEL")
        for got, want, tag in ((fast, ref, "step"), (fast_sense, ref_sense, "sense")):
            for k in ("next_agent", "next_hits") + (("reward", "cost") if tag == "step" else ()):
                np.testing.assert_array_equal(got[k].view(np.uint32), want[k].view(np.uint32), err_msg=f"{tag} {k}")
            _assert_graph_equal(got["graph"], want["graph"])
    assert np.isnan(fast["cost"][5, 0, 1]) and np.isfinite(fast["cost"][6]).all()
    # materialise-only mode
    from dgppo_amd import ops_env as O
    g1, g2 = O.alloc_graph(cfg, B, cuda), O.alloc_graph(cfg, B, cuda)
    h = fast["next_hits"]
    O.graph_materialize(cfg, _to(fast["next_agent"], cuda), _to(goal, cuda), _to(obst, cuda), _to(h, cuda), g1)
    monkeypatch.setenv("DGPPO_NO_WAVE_ENV_KERNEL", "1")
    O.graph_materialize(cfg, _to(fast["next_agent"], cuda), _to(goal, cuda), _to(obst, cuda), _to(h, cuda), g2)
    monkeypatch.delenv("DGPPO_NO_WAVE_ENV_KERNEL")
    torch.cuda.synchronize()
    _assert_graph_equal({k: v.cpu().numpy() for k, v in g1.items()}, {k: v.cpu().numpy() for k, v in g2.items()})
    _assert_graph_equal({k: v.cpu().numpy() for k, v in g1.items()}, fast["graph"])
    # and the oracle (double integrator: bit-exact; the bicycle's dynamics are compared elsewhere within 1e-6)
    if not ocfg.is_bicycle:
        want = E.env_step(ocfg, agent, goal, obst, hits, action, tab)
        for k in ("next_agent", "next_hits", "reward", "cost"):
            nan = np.isnan(want[k])                      # a NaN's sign bit is not specified: NaN == NaN, all else bitwise
            np.testing.assert_array_equal(np.isnan(fast[k]), nan, err_msg=k)
            np.testing.assert_array_equal(fast[k].view(np.uint32)[~nan], want[k].view(np.uint32)[~nan], err_msg=k)
        _assert_graph_equal(fast["graph"], want["graph"])
        # the corner cases were really there: an agent with all-zero alphas, and all-miss agents
        p = want["next_agent"][0, 0, :2]
        assert np.all(want["next_hits"][0, 0] == p), "agent inside an obstacle: all hit points collapse onto the agent"
        assert (np.abs(want["next_hits"]) > 1e5).any()


# ---- task variants -------------------------------------------------------------------------------------
VARIANTS = [("LidarLine", 4, 3), ("LidarLine", 6, 2), ("MPELine", 3, 3), ("MPELine", 5, 2), ("MPEFormation", 4, 3),
            ("MPECorridor", 4, 2), ("MPEConnectSpread", 4, 1), ("MPEConnectSpread", 6, 1)]


@pytest.mark.parametrize("kind,n,n_obs", VARIANTS)
def test_variant_step_matches_oracle(cuda, kind, n, n_obs):
    """LidarLine / MPELine (2 landmark nodes, goals on the segment), MPEFormation (1 landmark, goals on a circle), MPECorridor
    (obstacle edges always connected, y limit 2 A), MPEConnectSpread (third cost, clipped on both sides): the generic step
    kernel against the oracle — everything bit-exact except MPEFormation's reward (device cosf / sinf: 1e-6)."""
    cfg, ocfg = _mk(kind, n, n_obs)
    assert (cfg.n_goals, cfg.n_cost, cfg.n_obs) == (ocfg.n_goals, ocfg.n_cost, ocfg.n_obs)
    B = 48
    agent, goal, obst, action = _random_state(ocfg, B, seed=17 + n)
    if kind in ("MPECorridor", "MPEConnectSpread"):
        agent[::3, :, 1] += f32(0.9)                                  # some agents beyond y = area: only legal with y_limit = 2 A
        agent[::3, :, 3] = f32(1.0)
        agent[1::3, 0, :2] = f32(0.01)                                # one straggler: the team is disconnected in these envs
    tab = E.ray_table(32)
    hits = E.lidar_sense(ocfg, agent[..., :2], obst, *tab)[0] if ocfg.is_lidar else None
    want = E.env_step(ocfg, agent, goal, obst, hits, action, tab)
    got = _run_step(cfg, ocfg, agent, goal, obst, hits, action, cuda)
    np.testing.assert_array_equal(got["next_agent"].view(np.uint32), want["next_agent"].view(np.uint32))
    np.testing.assert_array_equal(got["cost"].view(np.uint32), want["cost"].view(np.uint32))
    assert got["cost"].shape == (B, n, ocfg.n_cost)
    if kind == "MPEFormation":
        np.testing.assert_allclose(got["reward"], want["reward"], atol=1e-7, rtol=1e-6)
    else:
        np.testing.assert_array_equal(got["reward"].view(np.uint32), want["reward"].view(np.uint32))
    if want["next_hits"] is not None:
        np.testing.assert_array_equal(got["next_hits"].view(np.uint32), want["next_hits"].view(np.uint32))
    _assert_graph_equal(got["graph"], want["graph"])
    assert got["graph"]["nodes"].shape == (B, ocfg.num_nodes, ocfg.node_dim)
    if kind in ("MPECorridor", "MPEConnectSpread"):
        assert (got["next_agent"][..., 1] > ocfg.area_size).any(), "the 2 A y-limit was never exercised"
        pad = ocfg.num_nodes - 1
        obs_block = want["graph"]["senders"][:, n * n + n * ocfg.n_goals:]
        assert (obs_block != pad).all(), "agent-obstacle edges are always connected in these two tasks"
    if kind == "MPEConnectSpread":
        c2 = got["cost"][..., 2]
        assert (c2 == c2[:, :1]).all() and (c2 > 0).any() and (c2 < 0).any()      # one value per env, both signs occur


@pytest.mark.parametrize("kind,n,n_obs", VARIANTS)
def test_variant_reset_matches_oracle_stream(cuda, kind, n, n_obs):
    from dgppo_amd import ops_env as O
    cfg, ocfg = _mk(kind, n, n_obs)
    B = 24
    seeds = np.arange(1, B + 1, dtype=np.int64) * 104729 + 7
    agent = torch.empty(B, n, 4, device=cuda)
    goal = torch.empty(B, cfg.n_goals, 4, device=cuda)
    obst = torch.empty(B, cfg.n_obs, cfg.obst_stride, device=cuda)
    O.env_reset(cfg, torch.from_numpy(seeds).to(cuda), agent, goal, obst)
    torch.cuda.synchronize()
    wa, wg, wo = E.env_reset(ocfg, seeds)
    np.testing.assert_array_equal(agent.cpu().numpy().view(np.uint32), wa.view(np.uint32))
    if kind == "MPEFormation":
        np.testing.assert_array_equal(goal.cpu().numpy().view(np.uint32), wg.view(np.uint32))
        # obstacle rejection compares against the circle goals (device cosf / sinf): identical decisions for these seeds
        np.testing.assert_allclose(obst.cpu().numpy(), wo, atol=0, rtol=0)
    elif kind == "LidarLine":
        np.testing.assert_array_equal(goal.cpu().numpy().view(np.uint32), wg.view(np.uint32))
        np.testing.assert_allclose(obst.cpu().numpy(), wo, atol=1e-6, rtol=0)      # rectangle corners: device cosf / sinf
    else:
        np.testing.assert_array_equal(goal.cpu().numpy().view(np.uint32), wg.view(np.uint32))
        np.testing.assert_array_equal(obst.cpu().numpy().view(np.uint32), wo.view(np.uint32))
