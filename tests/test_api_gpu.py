"""The drop-in Python surface (dgppo.env / dgppo.algo / dgppo.trainer, train.py) on the GPU: single-graph env API vs the
oracle, algo act/step/collect/update/save/load, lazy rollout.graph, Trainer loop and the train.py CLI."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import env_np as E

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _np(x):
    return x.detach().cpu().numpy()


@pytest.mark.parametrize("env_id,n,obs", [("LidarSpread", 4, 2), ("MPETarget", 3, 0), ("MPESpread", 3, 3), ("LidarTarget", 3, 1),
                                          # task variants: the goal NODES are 2 landmarks / 1 landmark / n goals
                                          ("LidarLine", 4, 2), ("MPELine", 5, 2), ("MPEFormation", 4, 3), ("MPECorridor", 3, 2),
                                          ("MPEConnectSpread", 4, 1)])
def test_single_graph_env_api_matches_oracle(cuda, env_id, n, obs):
    from dgppo.env import make_env
    env = make_env(env_id, n, num_obs=obs)
    ocfg = E.EnvCfg(E.KIND_NAMES[env_id], n_agents=n, n_obs=obs)
    assert (env.state_dim, env.node_dim, env.edge_dim, env.action_dim, env.n_cost) == (4, 7, 4, 2, ocfg.n_cost)
    assert env.max_episode_steps == 128 and env.dt == 0.03 and env.area_size == ocfg.area_size
    assert env.num_goals == ocfg.n_goals and len(env.cost_components) == ocfg.n_cost and env.params["n_obs"] == ocfg.n_obs
    g = env.reset(1234)
    N_, E_ = ocfg.num_nodes, ocfg.num_edges
    assert g.nodes.shape == (N_, 7) and g.edges.shape == (E_, 4) and g.states.shape == (N_, 4)
    assert g.receivers.shape == g.senders.shape == (E_,) and g.node_type.shape == (N_,)
    assert int(g.n_node) == N_ and int(g.n_edge) == E_ and g.is_single
    np.testing.assert_array_equal(_np(g.type_states(0, n)), _np(g.states[:n]))
    agent, goal = _np(g.type_states(0, n))[None], _np(g.type_states(1, env.num_goals))[None]
    if ocfg.is_lidar:
        rec = np.zeros((1, obs, 16), np.float32)
        ob = g.env_states.obstacle
        rec[0, :, 0:2], rec[0, :, 2], rec[0, :, 3], rec[0, :, 4] = _np(ob.center), _np(ob.width), _np(ob.height), _np(ob.theta)
        rec[0, :, 5], rec[0, :, 6] = np.cos(rec[0, :, 4]), np.sin(rec[0, :, 4])
        rec[0, :, 8:] = _np(ob.points).reshape(obs, 8)
        hits = _np(g.type_states(2, n * 8))[None, :, :2].reshape(1, n, 8, 2)
        obst = rec
    else:
        obst = _np(g.type_states(2, obs))[None] if obs > 0 else None
        hits = None
    action = np.random.default_rng(0).uniform(-1.3, 1.3, size=(n, 2)).astype(np.float32)
    res = env.step(g, torch.from_numpy(action))
    want = E.env_step(ocfg, agent, goal, obst, hits, action[None], E.ray_table(32))
    np.testing.assert_allclose(_np(res.reward), want["reward"][0], atol=1e-7)
    assert tuple(res.cost.shape) == (n, ocfg.n_cost)
    np.testing.assert_allclose(_np(res.cost), want["cost"][0], atol=1e-6)
    np.testing.assert_allclose(_np(res.graph.states[:n]), want["next_agent"][0], atol=1e-7)
    np.testing.assert_array_equal(_np(res.graph.receivers), want["graph"]["receivers"][0])
    np.testing.assert_array_equal(_np(res.graph.senders), want["graph"]["senders"][0])
    np.testing.assert_allclose(_np(res.graph.edges), want["graph"]["edges"][0], atol=1e-6)
    np.testing.assert_allclose(_np(env.get_cost(g)), want["cost"][0], atol=1e-6)
    assert not bool(res.done) and res.info == {}
    # env.get_graph(env_state[, lidar_data]) (lidar_env/base.py:227, mpe/base.py:211): rebuilds the same GraphsTuple from the
    # env state, with the hit points handed in and with the hit points sensed from the state
    es = res.graph.env_states
    variants = [env.get_graph(es)]
    if ocfg.is_lidar and obs > 0:
        k = 8
        lidar = res.graph.type_states(2, n * k)[:, :2]
        variants += [env.get_graph(es, lidar.reshape(n, k, 2)), env.get_graph(es, lidar)]
    for g2 in variants:
        for f in ("nodes", "edges", "states", "receivers", "senders", "node_type"):
            assert torch.equal(getattr(g2, f), getattr(res.graph, f)), f
        assert int(g2.n_node) == N_ and int(g2.n_edge) == E_


def _mk_algo(env, batch_size, seed=0, train_steps=100):
    from dgppo.algo import make_algo
    return make_algo(algo="dgppo", env=env, node_dim=env.node_dim, edge_dim=env.edge_dim, state_dim=env.state_dim,
                     action_dim=env.action_dim, n_agents=env.num_agents, cost_weight=0.0, cbf_weight=1.0, actor_gnn_layers=2,
                     Vl_gnn_layers=2, Vh_gnn_layers=1, rnn_layers=1, lr_actor=3e-4, lr_Vl=1e-3, lr_Vh=1e-3, max_grad_norm=2.0,
                     alpha=10.0, cbf_eps=1e-2, seed=seed, batch_size=batch_size, use_rnn=True, use_lstm=False, coef_ent=1e-2,
                     rnn_step=16, gamma=0.99, clip_eps=0.25, lagr_init=0.5, lr_lagr=1e-7, train_steps=train_steps,
                     cbf_schedule=True, cost_schedule=False)


def test_algo_surface_collect_update_save_load(cuda, tmp_path):
    from dgppo.env import make_env
    env = make_env("LidarSpread", 3, max_step=32, num_obs=2)
    algo = _mk_algo(env, batch_size=8 * 32)
    assert algo.init_rnn_state.shape == (1, 3, 1, 64) and float(algo.init_rnn_state.abs().max()) == 0
    cfgd = algo.config
    for k in ("gamma", "lr_actor", "lr_Vl", "lr_Vh", "batch_size", "clip_eps", "gae_lambda", "coef_ent", "rnn_step", "alpha",
              "cbf_eps", "cbf_weight", "cbf_schedule", "Vh_gnn_layers"):
        assert k in cfgd
    p = algo.params
    assert set(p) == {"policy", "Vl", "Vh"} and "PolicyNet_0" in p["policy"]["params"]
    g = env.reset(7)
    a, h = algo.act(g, algo.init_rnn_state)
    assert a.shape == (3, 2) and h.shape == (1, 3, 1, 64) and float(a.abs().max()) <= 1
    a2, lp, h2 = algo.step(g, algo.init_rnn_state, 99)
    assert a2.shape == (3, 2) and lp.shape == (3,) and torch.allclose(h, h2)
    keys = np.arange(1, 17)
    ro = algo.collect(None, keys)
    B, T = 16, 32
    assert ro.actions.shape == (B, T, 3, 2) and ro.log_pis.shape == (B, T, 3) and ro.rewards.shape == (B, T)
    assert ro.costs.shape == (B, T, 3, 2) and ro.rnn_states.shape == (B, T, 1, 3, 1, 64) and ro.dones.shape == (B, T)
    assert ro.length == B and ro.time_horizon == T and ro.n_data == B * T
    # lazy graphs: [B, T, N, ...], next_graph[t] == graph[t+1], contents equal the oracle's get_graph
    ocfg = E.EnvCfg(E.LIDAR_SPREAD, n_agents=3, n_obs=2)
    assert ro.graph.nodes.shape == (B, T, ocfg.num_nodes, 7) and ro.next_graph.senders.shape == (B, T, ocfg.num_edges)
    np.testing.assert_array_equal(_np(ro.graph.nodes[:, 1:]), _np(ro.next_graph.nodes[:, :-1]))
    st = ro.graph.states
    hits = _np(st[:, 5, 6:6 + 24, :2]).reshape(B, 3, 8, 2)
    gg = E.get_graph(ocfg, _np(st[:, 5, :3]), _np(st[:, 5, 3:6]), None, hits)
    np.testing.assert_array_equal(_np(ro.graph.receivers[:, 5]), gg["receivers"])
    np.testing.assert_array_equal(_np(ro.graph.edges[:, 5]), gg["edges"])
    before = algo.params["policy"]["params"]["OutputDenseMean"]["kernel"].copy()
    info = algo.update(ro, 3)
    assert info["policy/has_nan"] == 0 and np.isfinite(info["policy/loss"])
    after = algo.params["policy"]["params"]["OutputDenseMean"]["kernel"]
    assert not np.array_equal(before, after)
    algo.save(str(tmp_path), 3)
    assert sorted(os.listdir(tmp_path / "3")) == ["Vh.pkl", "Vl.pkl", "actor.pkl"]
    algo2 = _mk_algo(env, batch_size=8 * 32, seed=5)
    algo2.load(str(tmp_path), 3)
    np.testing.assert_array_equal(algo2.params["policy"]["params"]["OutputDenseMean"]["kernel"], after)
    a3, _ = algo2.act(g, algo2.init_rnn_state)
    a4, _ = algo.act(g, algo.init_rnn_state)
    assert torch.equal(a3, a4)


def test_trainer_loop_and_metrics(cuda, tmp_path):
    from dgppo.env import make_env
    from dgppo.trainer.trainer import Trainer
    env, env_test = make_env("MPETarget", 3, max_step=16, num_obs=0), make_env("MPETarget", 3, max_step=16, num_obs=0)
    algo = _mk_algo(env, batch_size=16 * 16, train_steps=2)
    tr = Trainer(env=env, env_test=env_test, algo=algo, gamma=0.99, n_env_train=32, n_env_test=8, log_dir=str(tmp_path / "run"),
                 seed=0, params={"run_name": "t", "training_steps": 2, "eval_interval": 1, "eval_epi": 1, "save_interval": 2})
    tr.train()
    assert tr.update_steps == 3                                      # steps + 1 iterations (trainer.py:103)
    rows = [json.loads(l) for l in open(tmp_path / "run" / "metrics.jsonl")]
    keys = set().union(*[set(r) for r in rows])
    for k in ("eval/reward", "eval/reward_final", "eval/cost", "eval/unsafe_frac", "Vl/loss", "Vh/loss_Vh", "policy/loss",
              "policy/clip_frac", "policy/entropy", "policy/total_variation_dist", "eval/safe_data"):
        assert k in keys, k
    assert sorted(os.listdir(tmp_path / "run" / "models")) == ["0", "2"]


@pytest.mark.parametrize("env_id,n,obs", [("LidarLine", 4, 2), ("MPELine", 3, 3), ("MPEFormation", 4, 3), ("MPECorridor", 3, 2),
                                          ("MPEConnectSpread", 4, 1)])
def test_variant_envs_train_through_the_algo_surface(cuda, env_id, n, obs):
    """make_env -> make_algo -> collect -> update -> deterministic evaluation on every task variant:
    two DGPPO iterations with finite losses, the constraint-value net sized by env.n_cost."""
    from dgppo.env import make_env
    env = make_env(env_id, n, max_step=16, num_obs=obs)
    algo = _mk_algo(env, batch_size=4 * 16)
    assert algo.engine.Vh.n_out == env.n_cost
    for it in range(2):
        ro = algo.collect(None, np.arange(8, dtype=np.int64) + 100 * it + 1)
        assert tuple(ro.costs.shape) == (8, 16, n, env.n_cost) and tuple(ro.graph.states.shape[:2]) == (8, 16)
        info = algo.update(ro, it)
        assert all(np.isfinite(v) for v in info.values()), info
    ev = algo.collect_deterministic(np.arange(4, dtype=np.int64) + 7, env=env)
    assert torch.isfinite(ev.rewards).all() and tuple(ev.costs.shape) == (4, 16, n, env.n_cost)


def test_train_py_cli(cuda, tmp_path):
    cmd = [sys.executable, os.path.join(ROOT, "train.py"), "--env", "LidarSpread", "-n", "3", "--algo", "dgppo", "--obs", "1",
           "--steps", "1", "--n-env-train", "16", "--batch-size", "2048", "--n-env-test", "4", "--eval-interval", "1",
           "--save-interval", "1", "--log-dir", str(tmp_path / "logs")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "step:   0" in out.stdout and "unsafe_frac" in out.stdout
    runs = os.listdir(tmp_path / "logs" / "LidarSpread" / "dgppo")
    assert len(runs) == 1 and os.path.exists(tmp_path / "logs" / "LidarSpread" / "dgppo" / runs[0] / "config.yaml")


def test_test_py_cli_after_train(cuda, tmp_path):
    """train.py -> test.py round trip: config.yaml + models/{step}/*.pkl are read back, the printed /
    logged statistics equal the reductions recomputed from a deterministic rollout of the same checkpoint and seeds."""
    cmd = [sys.executable, os.path.join(ROOT, "train.py"), "--env", "LidarSpread", "-n", "3", "--algo", "dgppo", "--obs", "1",
           "--steps", "1", "--n-env-train", "16", "--batch-size", "2048", "--n-env-test", "4", "--eval-interval", "1",
           "--save-interval", "1", "--log-dir", str(tmp_path / "logs")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    run_dir = tmp_path / "logs" / "LidarSpread" / "dgppo"
    run_dir = run_dir / os.listdir(run_dir)[0]
    assert sorted(os.listdir(run_dir / "models")) == ["0", "1"]
    assert sorted(os.listdir(run_dir / "models" / "1")) == ["Vh.pkl", "Vl.pkl", "actor.pkl"]
    tcmd = [sys.executable, os.path.join(ROOT, "test.py"), "--path", str(run_dir), "--epi", "6", "--offset", "1", "--no-video",
            "--log", "--max-step", "32"]
    tout = subprocess.run(tcmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert tout.returncode == 0, tout.stderr[-2000:]
    assert "step:  1" in tout.stdout and "epi: 4, reward:" in tout.stdout and "epi: 5" not in tout.stdout
    # recompute in-process from the same checkpoint and seeds
    from dgppo.algo import make_algo
    from dgppo.env import make_env
    from dgppo_amd.trainer import evaluate as EV
    cfg = EV.load_config(str(run_dir / "config.yaml"))
    env = make_env(cfg.env, cfg.num_agents, num_obs=cfg.obs, max_step=32)
    algo = make_algo(algo=cfg.algo, env=env, node_dim=env.node_dim, edge_dim=env.edge_dim, state_dim=env.state_dim,
                     action_dim=env.action_dim, n_agents=env.num_agents, cost_weight=cfg.cost_weight,
                     actor_gnn_layers=cfg.actor_gnn_layers, Vl_gnn_layers=cfg.Vl_gnn_layers, Vh_gnn_layers=cfg.Vh_gnn_layers,
                     lr_actor=cfg.lr_actor, lr_Vl=cfg.lr_Vl, seed=cfg.seed, use_rnn=cfg.use_rnn, rnn_layers=cfg.rnn_layers,
                     use_lstm=cfg.use_lstm)
    algo.load(str(run_dir / "models"), 1)
    keys = np.random.default_rng([1234, 13]).integers(1, 2 ** 62, size=1000)[:6][1:]
    ro = algo.collect_deterministic(keys, env=env)
    agg = EV.aggregate(EV.episode_stats(_np(ro.rewards), _np(ro.costs)))
    last = [l for l in tout.stdout.splitlines() if l.startswith("reward:")][-1]
    assert f"reward: {agg['reward']:.3f}, min/max reward: {agg['reward_min']:.3f}/{agg['reward_max']:.3f}" in last
    assert f"safe_rate: {agg['safe_mean'] * 100:.3f}%" in last
    rows = open(run_dir / "test_log.csv").read().splitlines()
    assert rows == [EV.csv_line(env, 6, agg).strip()]
    # the stochastic variant runs and reports the same number of episodes
    sout = subprocess.run(tcmd[:-4] + ["--stochastic", "--max-step", "32", "--no-video"], capture_output=True, text=True,
                          timeout=600, cwd=ROOT)
    assert sout.returncode == 0 and "epi: 4, reward:" in sout.stdout, sout.stderr[-2000:]
    # without --no-video every episode is rendered (test.py:150-159): one animation per episode under videos/{step}/, built
    # from the GraphsTuples the materialise kernel emits for the stored rollout
    vcmd = [sys.executable, os.path.join(ROOT, "test.py"), "--path", str(run_dir), "--epi", "2", "--max-step", "6", "--dpi", "30"]
    vout = subprocess.run(vcmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert vout.returncode == 0, vout.stderr[-2000:]
    vids = sorted(os.listdir(run_dir / "videos" / "1"))
    assert len(vids) == 2 and all("_n3_epi0" in v and v.endswith((".gif", ".mp4")) for v in vids), vids
    if vids[0].endswith(".gif"):
        from PIL import Image, ImageSequence
        with Image.open(run_dir / "videos" / "1" / vids[0]) as im:
            assert sum(1 for _ in ImageSequence.Iterator(im)) == 6


def test_env_render_video_mpe(cuda, tmp_path):
    """env.render_video on a batched MPE rollout straight from the engine (disc obstacles, index= picks the episode)."""
    from dgppo.algo import make_algo
    from dgppo.env import make_env
    env = make_env("MPESpread", 3, num_obs=2, max_step=4)
    algo = make_algo(algo="dgppo", env=env, node_dim=env.node_dim, edge_dim=env.edge_dim, state_dim=env.state_dim,
                     action_dim=env.action_dim, n_agents=env.num_agents, n_env_train=2, batch_size=8, seed=0)
    ro = algo.collect_deterministic(np.array([5, 6], dtype=np.int64), env=env)
    out = env.render_video(ro, tmp_path / "mpe.gif", None, {}, dpi=30, index=1)
    assert out.exists() and out.stat().st_size > 500


def test_informarl_algo_round_trip(cuda, tmp_path):
    """make_algo("informarl"): collect / update / save / load / act through the reference's surface (informarl.py)."""
    from dgppo.algo import make_algo
    from dgppo.env import make_env
    env = make_env("LidarSpread", 3, num_obs=1, max_step=16)
    algo = make_algo(algo="informarl", env=env, node_dim=env.node_dim, edge_dim=env.edge_dim, state_dim=env.state_dim,
                     action_dim=env.action_dim, n_agents=env.num_agents, cost_weight=0.5, batch_size=128, rnn_step=8,
                     train_steps=10, seed=1)
    assert set(algo.params) == {"policy", "Vl"} and algo.config["cost_weight"] == 0.5 and "Vh_gnn_layers" not in algo.config
    keys = np.arange(1, 17)
    before = algo.engine.policy.params.clone()
    for step in range(2):
        info = algo.update(algo.collect(None, keys + step), step)
    assert all(np.isfinite(v) for v in info.values()) and not torch.equal(before, algo.engine.policy.params)
    assert "Vh/loss_Vh" not in info and "Vl/loss" in info
    algo.save(str(tmp_path), 7)
    assert sorted(os.listdir(tmp_path / "7")) == ["Vl.pkl", "actor.pkl"]
    algo2 = make_algo(algo="informarl", env=env, node_dim=env.node_dim, edge_dim=env.edge_dim, state_dim=env.state_dim,
                      action_dim=env.action_dim, n_agents=env.num_agents, batch_size=128, rnn_step=8, train_steps=10, seed=2)
    algo2.load(str(tmp_path), 7)
    assert torch.equal(algo2.engine.policy.params, algo.engine.policy.params)
    g = env.reset(5)
    a1, _ = algo.act(g, algo.init_rnn_state)
    a2, _ = algo2.act(g, algo2.init_rnn_state)
    assert torch.equal(a1, a2)


def test_hcbfcrpo_algo_runs(cuda):
    """make_algo("hcbfcrpo") through the reference's surface: two collect/update iterations, DGPPO's config keys."""
    from dgppo.algo import make_algo
    from dgppo.env import make_env
    env = make_env("MPETarget", 3, num_obs=2, max_step=16)
    algo = make_algo(algo="hcbfcrpo", env=env, node_dim=env.node_dim, edge_dim=env.edge_dim, state_dim=env.state_dim,
                     action_dim=env.action_dim, n_agents=env.num_agents, batch_size=128, rnn_step=8, train_steps=10, seed=1)
    assert set(algo.params) == {"policy", "Vl"} and "cbf_weight" in algo.config and "alpha" in algo.config
    for step in range(2):
        info = algo.update(algo.collect(None, np.arange(1, 17) + step), step)
    assert all(np.isfinite(v) for v in info.values()) and "eval/safe_data" in info and "Vh/loss_Vh" not in info


def test_collect_then_evaluate_then_update_is_safe(cuda):
    """collect() prepares the deterministic rollout update() needs in engine-owned buffers; a deterministic evaluation of the
    same batch size in between reuses those buffers, so the hand-over must be dropped and update() must still work."""
    from dgppo.algo import make_algo
    from dgppo.env import make_env
    env = make_env("LidarSpread", 3, num_obs=1, max_step=16)
    algo = make_algo(algo="dgppo", env=env, node_dim=env.node_dim, edge_dim=env.edge_dim, state_dim=env.state_dim,
                     action_dim=env.action_dim, n_agents=env.num_agents, batch_size=128, rnn_step=8, train_steps=10, seed=3)
    keys = np.arange(1, 17)
    ro = algo.collect(None, keys)
    assert algo._pending_det is not None
    algo.collect_deterministic(keys + 100, env=env)          # same batch size: reuses the (16, deterministic) buffers
    assert algo._pending_det is None
    info = algo.update(ro, 0)
    assert all(np.isfinite(v) for v in info.values())


@pytest.mark.parametrize("env_id,n,obs", [("LidarSpread", 3, 2), ("MPESpread", 3, 3)])
def test_reference_signature_rollout_functions(cuda, env_id, n, obs):
    """dgppo.trainer.utils.rollout / test_rollout with the reference's (env, actor, init_rnn_state, key) signatures
    (dgppo/trainer/utils.py:22,60): one environment stepped through env.reset / env.step and the algo's single-graph
    act / step.  The deterministic one must reproduce the batched engine rollout of the same scene exactly, including
    the post-step carry convention; the stochastic one must store the pre-step carry and consistent log-probabilities."""
    import functools as ft
    from dgppo.env import make_env
    from dgppo.trainer import utils as TU
    from dgppo.trainer.data import Rollout
    T_ = 6
    env = make_env(env_id, n, max_step=T_, num_obs=obs)
    algo = _mk_algo(env, batch_size=4 * T_)
    key = 4242
    r = TU.test_rollout(env, ft.partial(algo.act, params=None), algo.init_rnn_state, key)
    assert isinstance(r, Rollout) and r.log_pis is None
    assert r.actions.shape == (T_, n, 2) and r.rewards.shape == (T_,) and r.costs.shape == (T_, n, 2)
    assert r.rnn_states.shape == (T_, 1, n, 1, 64) and r.graph.nodes.shape[0] == T_ and r.dones.shape == (T_,)
    np.testing.assert_array_equal(_np(r.graph.nodes[1:]), _np(r.next_graph.nodes[:-1]))
    key_x0 = TU._split(key, 2)[0]
    b = algo.collect_deterministic(np.array([key_x0], dtype=np.int64))
    # same kernels at B = 1; a tolerance (not bit equality) because launch geometry may differ between the two paths
    tol = dict(atol=5e-6, rtol=0)
    np.testing.assert_allclose(_np(r.actions), _np(b.actions[0]), **tol)
    np.testing.assert_allclose(_np(r.rewards), _np(b.rewards[0]), **tol)
    np.testing.assert_allclose(_np(r.costs), _np(b.costs[0]), **tol)
    np.testing.assert_allclose(_np(r.rnn_states), _np(b.rnn_states[0]), **tol)          # post-step carry (utils.py:71-77)
    np.testing.assert_allclose(_np(r.graph.states[:, :2 * n]), _np(b.graph.states[0][:, :2 * n]), **tol)
    # stochastic: pre-step carry stored (utils.py:46-51) — rnn_states[0] is the initial carry, rnn_states[t+1] is what the
    # actor returned at step t; log_pi is the actor's own
    s = TU.rollout(env, ft.partial(algo.step, params=None), algo.init_rnn_state, key)
    assert s.log_pis.shape == (T_, n) and s.rnn_states.shape == (T_, 1, n, 1, 64)
    assert float(s.rnn_states[0].abs().max()) == 0.0
    k_x0, _, k_steps = TU._split(key, 3)
    _, lp1, h1 = algo.step(env.reset(k_x0), algo.init_rnn_state, TU._split(k_steps, T_)[0])
    np.testing.assert_array_equal(_np(s.log_pis[0]), _np(lp1))
    np.testing.assert_array_equal(_np(s.rnn_states[1]), _np(h1))
    assert float(s.actions.abs().max()) <= 1.0 and bool(torch.isfinite(s.log_pis).all())


def test_informarl_lagr_algo_round_trip(cuda, tmp_path):
    """make_algo("informarl_lagr") through the reference's surface (informarl_lagr.py:25-327): three networks, the
    multipliers move away from lagr_init and stay >= 0, checkpoints {actor,Vl,Vh}.pkl round-trip (the 128-wide head too)."""
    from dgppo.algo import make_algo
    from dgppo.env import make_env
    env = make_env("LidarSpread", 3, num_obs=1, max_step=16)
    mk = lambda seed: make_algo(algo="informarl_lagr", env=env, node_dim=env.node_dim, edge_dim=env.edge_dim,
                                state_dim=env.state_dim, action_dim=env.action_dim, n_agents=env.num_agents, batch_size=128,
                                rnn_step=8, train_steps=10, seed=seed, lagr_init=0.5, lr_lagr=1e-2, Vh_gnn_layers=1)
    algo = mk(1)
    assert set(algo.params) == {"policy", "Vl", "Vh"}
    assert algo.params["Vh"]["params"]["ValueGNNHead"]["Dense_0"]["kernel"].shape == (128, 64)
    for k in ("lr_Vh", "Vh_gnn_layers", "lagr_init", "lr_lagr", "cost_weight"):
        assert k in algo.config
    assert algo.ah_lagr.shape == (3, 2) and float(algo.ah_lagr.min()) == 0.5 and algo.init_Vh_rnn_state.shape == (1, 3, 1, 64)
    for step in range(3):                      # the third update replays the captured minibatch graph
        info = algo.update(algo.collect(None, np.arange(1, 17) + step), step)
    assert all(np.isfinite(v) for v in info.values())
    for k in ("Vh/loss", "Vh/grad_norm", "Vh/has_nan", "Vh/max_target", "Vh/min_target", "policy/lagr_mean", "Vl/loss"):
        assert k in info
    lg = algo.ah_lagr.cpu().numpy()
    assert (lg >= 0).all() and not np.allclose(lg, 0.5)
    algo.save(str(tmp_path), 3)
    assert sorted(os.listdir(tmp_path / "3")) == ["Vh.pkl", "Vl.pkl", "actor.pkl"]
    algo2 = mk(2)
    algo2.load(str(tmp_path), 3)
    for k in ("policy", "Vl", "Vh"):
        assert torch.equal(algo2.engine.nets[k].params, algo.engine.nets[k].params)


def test_epoch_ppo_runs_that_many_passes(cuda):
    """epoch_ppo > 1 (dgppo.py:154-172): every epoch is a full pass over reshuffled minibatches with recomputed targets —
    the optimisers must have stepped epoch_ppo * n_minibatches times."""
    from dgppo.env import make_env
    env = make_env("MPESpread", 3, max_step=16, num_obs=2)
    from dgppo.algo import make_algo
    algo = make_algo(algo="dgppo", env=env, node_dim=env.node_dim, edge_dim=env.edge_dim, state_dim=env.state_dim,
                     action_dim=env.action_dim, n_agents=env.num_agents, batch_size=4 * 16, rnn_step=8, train_steps=10, seed=1,
                     epoch_ppo=3)
    assert algo.config["epoch_ppo"] == 3
    info = algo.update(algo.collect(None, np.arange(1, 17)), 0)
    assert all(np.isfinite(v) for v in info.values())
    for k in ("policy", "Vl", "Vh"):
        assert float(algo.engine.opt[k].state[2]) == 3 * (16 // 4)


@pytest.mark.parametrize("flags,shape", [(dict(use_rnn=False), (1, 3, 1, 64)), (dict(rnn_layers=2), (2, 3, 1, 64)),
                                         (dict(use_lstm=True), (1, 3, 2, 64)), (dict(use_lstm=True, rnn_layers=2), (2, 3, 2, 64))])
def test_rnn_option_flags_through_the_algo_surface(cuda, flags, shape):
    """--no-rnn / --rnn-layers / --use-lstm (train.py:30-33): carries have the reference's (n_layers, n_agents, n_carries,
    64) shape at the API (n_carries = 2 for the LSTM's (c, h)), act / collect / update run."""
    from dgppo.algo import make_algo
    from dgppo.env import make_env
    env = make_env("LidarSpread", 3, num_obs=1, max_step=16)
    kw = dict(env=env, node_dim=env.node_dim, edge_dim=env.edge_dim, state_dim=env.state_dim, action_dim=env.action_dim,
              n_agents=env.num_agents, batch_size=128, rnn_step=8, train_steps=10, seed=1)
    algo = make_algo(algo="dgppo", **kw, **flags)
    assert algo.init_rnn_state.shape == shape
    g = env.reset(3)
    a, h = algo.act(g, algo.init_rnn_state)
    assert a.shape == (3, 2) and h.shape == shape
    if flags.get("use_rnn", True):
        a2, h2 = algo.act(g, h)
        assert not torch.equal(h, h2)
    else:
        assert float(h.abs().max()) == 0.0
    ro = algo.collect(None, np.arange(1, 17))
    assert ro.rnn_states.shape == (16, 16) + shape
    info = algo.update(ro, 0)
    assert all(np.isfinite(v) for v in info.values())
