"""End-to-end GPU parity of the DGPPO engine (rollout -> value pre-passes -> GAE -> advantage -> minibatch gradients)
against the torch/numpy oracle, on small configurations; plus rollout-level checks."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import dgppo_ref as R
from oracle import env_np as E
from oracle import nn_torch as T

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _np_rollout(ro):
    c = lambda x: None if x is None else x.detach().cpu().numpy()
    return dict(agent=c(ro.agent), hits=c(ro.hits), goal=c(ro.goal), obst=c(ro.obst), actions=c(ro.actions),
                log_pis=c(ro.log_pis), rnn_states=c(ro.rnn_states.contiguous()), rewards=c(ro.rewards), costs=c(ro.costs))


def _setup(kind_name, n, n_obs, B, T_, cuda, batch_size, rnn_step, use_rnn=True, rnn_layers=1, use_lstm=False, **engine_kw):
    from dgppo_amd import _native as N, engine as EN, init
    kind = N.ENV_KINDS[kind_name]
    cfg = N.make_env_cfg(kind, n, n_obs)
    ocfg = E.EnvCfg(kind, n_agents=n, n_obs=n_obs)
    hp = EN.Hyper(batch_size=batch_size, rnn_step=rnn_step, train_steps=100, use_rnn=use_rnn, rnn_layers=rnn_layers,
                  use_lstm=use_lstm)
    eng = EN.Engine(cfg, hp, cuda, T=T_, **engine_kw)
    nc = rnn_layers if use_rnn else 0
    trees = {"policy": init.init_policy(0, cfg.node_dim, 2, 2, nc, use_lstm),
             "Vl": init.init_value(0, cfg.node_dim, 1, 2, 2, rnn_layers=nc, lstm=use_lstm),
             "Vh": init.init_value(0, cfg.node_dim, cfg.n_cost, 1, 3, rnn_layers=min(nc, 1))}
    rng = np.random.default_rng(11)
    jitter = lambda tr: T.tree_map(lambda a: torch.from_numpy(a + 0.05 * rng.standard_normal(a.shape).astype(np.float32)), tr)
    trees = {k: jitter(v) for k, v in trees.items()}
    trees["policy"]["params"]["ScaleHid"]["kernel"] = T.orthogonal(torch.Generator().manual_seed(1), 64, 64, 0.5)
    for k, net in eng.nets.items():
        net.load_tree(trees[k])
    eng.set_entropy_noise(77)
    return cfg, ocfg, hp, eng, trees


def _close(got, want, name, tol=1e-5):
    """north_star's fp32 bar: |got - want| <= 1e-5 of the output scale"""
    got = got.detach().cpu().numpy() if torch.is_tensor(got) else np.asarray(got)
    scale = max(1.0, float(np.abs(want).max()))
    err = float(np.abs(got - want).max())
    assert err <= tol * scale, f"{name}: max error {err:.3e} > {tol:g} x scale {scale:.3g}"


def _check_advantage(tg, wt, dt, alpha, cbf_eps, w, label=""):
    """The advantage merge has a HARD gate (safe = all_h(cdot <= 0), dgppo.py:246-259) on cdot = (Vh[t+1]-Vh[t])/dt + alpha*Vh[t],
    which amplifies any fp32 difference in Vh by 1/dt + alpha ~ 43.  Three separate statements instead of one loose one:
      (1) the advantage KERNEL on the device's own Vl / Vh / Ql equals the oracle formula on those same inputs to 1e-5,
          gate flips allowed only where |cdot| is at rounding level (< 1e-5) and counted;
      (2) end to end, a gate differs from the oracle's only where the oracle's own |cdot| is inside the propagated error
          band of Vh (a borderline decision), and such flips are rare;
      (3) end to end, every entry whose gate agrees is within the propagated numeric bound."""
    from oracle import algo_ref as A
    g = {k: tg[k].cpu().numpy() for k in ("Vl", "Vh", "Ql", "adv")}
    # (1) same inputs
    same_in, _ = A.advantage(g["Ql"], g["Vl"], g["Vh"], dt, alpha, cbf_eps, w)
    deriv_g = (g["Vh"][:, 1:] - g["Vh"][:, :-1]) / np.float32(dt) + np.float32(alpha) * g["Vh"][:, :-1]
    rounding_level = (np.abs(deriv_g) < 1e-5).any(axis=-1)
    bad1 = np.abs(g["adv"] - same_in) > 1e-5 * np.maximum(1, np.abs(same_in))
    assert not (bad1 & ~rounding_level).any(), f"{label}: advantage kernel off by more than 1e-5 on identical inputs"
    # (2) gate decisions end to end
    errVh = float(np.abs(g["Vh"] - wt["Vh"]).max())
    band = errVh * (2.0 / dt + alpha) + 1e-6
    deriv_o = (wt["Vh"][:, 1:] - wt["Vh"][:, :-1]) / np.float32(dt) + np.float32(alpha) * wt["Vh"][:, :-1]
    safe_g, safe_o = (deriv_g <= 0).all(axis=-1), (deriv_o <= 0).all(axis=-1)
    flipped = safe_g != safe_o
    borderline = (np.abs(deriv_o) <= band).any(axis=-1)
    assert not (flipped & ~borderline).any(), f"{label}: a safe-gate decision differs away from the threshold (band {band:.2e})"
    assert flipped.mean() < 0.02, f"{label}: {flipped.sum()} of {flipped.size} gates flipped"
    # (3) numeric error where the gate agrees: d(adv) <= d(Al) + w * d(cdot)
    Al_o = wt["Ql"] - wt["Vl"][:, :-1]
    std = Al_o.std(axis=1, keepdims=True) + 1e-8
    errAl = (np.abs(g["Ql"] - wt["Ql"]).max() + np.abs(g["Vl"] - wt["Vl"]).max()) * 4.0 / std
    bound = errAl[:, :, None] + w * band + 1e-5
    err = np.abs(g["adv"] - wt["adv"])
    assert (err[~flipped] <= np.broadcast_to(bound, err.shape)[~flipped]).all(), \
        f"{label}: advantage error {err[~flipped].max():.2e} exceeds the propagated bound"
    return int(flipped.sum())


def _check_first_minibatch_grads(leaf, grads, names, tol=5e-5):
    for name in names:
        w = dict(T.tree_leaves(T.tree_map(lambda t: t.grad if t.grad is not None else torch.zeros_like(t), leaf[name])))
        gt = dict(T.tree_leaves(T.tree_map(lambda a: torch.from_numpy(np.ascontiguousarray(a)), grads[name])))
        scale = max(float(v.abs().max()) for v in w.values())
        for k in w:
            err = float((gt[k].double() - w[k].double()).abs().max())
            assert err <= tol * max(scale, 1e-3), f"{name} grad {k}: err {err:.3e} scale {scale:.3e}"


@pytest.mark.parametrize("kind,n,n_obs", [("LidarSpread", 3, 2), ("MPETarget", 3, 0)])
def test_rollout_matches_oracle_stepwise(cuda, kind, n, n_obs):
    """stochastic + deterministic rollouts: every stored quantity re-derived by the oracle from the same noise."""
    B, T_ = 6, 8
    cfg, ocfg, hp, eng, trees = _setup(kind, n, n_obs, B, T_, cuda, 16, 4)
    seeds = torch.arange(1, B + 1, dtype=torch.int64, device=cuda) * 104729
    for stochastic in (True, False):
        ro = eng.rollout(seeds, stochastic, noise_seed=5).finalize()
        r = _np_rollout(ro)
        wa, wg, wo = E.env_reset(ocfg, [int(s) for s in seeds.cpu().numpy()])
        np.testing.assert_array_equal(r["agent"][:, 0], wa)
        np.testing.assert_array_equal(r["goal"], wg)
        if wo is not None and r["obst"] is not None:   # trig-derived rectangle fields differ by <= 1 ulp (device cos/sin):
            np.testing.assert_allclose(r["obst"], wo, atol=1e-6)
            wo = r["obst"]                       # continue with the device's own records
        tab = E.ray_table(32)
        hits = E.lidar_sense(ocfg, wa[..., :2], wo, *tab)[0] if (ocfg.is_lidar and n_obs > 0) else None
        eps = eng.arena.get("ro.eps", T_, B * n, 2).cpu().numpy().reshape(T_, B, n, 2) if stochastic else None
        h = torch.zeros(B, n, 64)
        agent = wa
        for t in range(T_):
            g = T.graph_to_torch(E.get_graph(ocfg, agent, wg, wo, hits))
            with torch.no_grad():
                if stochastic:
                    a, lp, h_new = T.policy_sample(trees["policy"], g, h, n, torch.from_numpy(eps[t]))
                    _close(r["log_pis"][:, t], lp.numpy(), "log_pi")
                else:
                    a, h_new = T.policy_mode(trees["policy"], g, h, n)
            np.testing.assert_allclose(r["actions"][:, t], a.numpy(), atol=1e-5)
            stored = h if stochastic else h_new
            np.testing.assert_allclose(r["rnn_states"][:, t], stored.numpy(), atol=1e-5)
            # feed the DEVICE action to the oracle env so trajectories cannot drift apart
            out = E.env_step(ocfg, r["agent"][:, t], wg, wo, r["hits"][:, t] if hits is not None else None, r["actions"][:, t], tab)
            np.testing.assert_array_equal(r["agent"][:, t + 1], out["next_agent"])
            np.testing.assert_array_equal(r["rewards"][:, t], out["reward"])
            np.testing.assert_array_equal(r["costs"][:, t], out["cost"])
            agent, hits, h = out["next_agent"], out["next_hits"], h_new
            if hits is not None:
                np.testing.assert_array_equal(r["hits"][:, t + 1], hits)


@pytest.mark.parametrize("kind,n,n_obs", [("LidarSpread", 3, 2), ("MPESpread", 3, 3),
                                           # task variants: 2 / 1 landmark nodes, and n_cost = 3
                                           ("LidarLine", 4, 2), ("MPEFormation", 4, 3), ("MPEConnectSpread", 4, 1),
                                           ("MPECorridor", 3, 2), ("MPELine", 3, 2)])
def test_update_targets_and_gradients(cuda, kind, n, n_obs):
    B, T_, rs, bs = 4, 8, 4, 16
    cfg, ocfg, hp, eng, trees = _setup(kind, n, n_obs, B, T_, cuda, bs, rs)
    seeds = torch.arange(1, B + 1, dtype=torch.int64, device=cuda) * 7919
    ro = eng.rollout(seeds, True, noise_seed=3)
    det = eng.rollout(seeds + 1000, False)
    ro.finalize(); det.finalize()
    step = 60                                                       # past 50 % of train_steps: schedule weight x2
    tg = eng.targets(ro, det, step)
    hpd = dict(gamma=hp.gamma, gae_lambda=hp.gae_lambda, alpha=hp.alpha, cbf_eps=hp.cbf_eps, rnn_step=rs,
               clip_eps=hp.clip_eps, coef_ent=hp.coef_ent)
    r, d = _np_rollout(ro), _np_rollout(det)
    leaf = {k: T.tree_map(lambda t: t.clone().requires_grad_(), v) for k, v in trees.items()}
    wt = R.targets(leaf, ocfg, r, d, hpd, eng.cbf_weight_at(step))
    assert eng.cbf_weight_at(step) == 2.0
    for k in ("Vl", "Vh", "Vh_det", "Ql", "Qh", "Qh_det"):
        _close(tg[k], wt[k], k)
    _check_advantage(tg, wt, ocfg.dt, hp.alpha, hp.cbf_eps, eng.cbf_weight_at(step), kind)
    # gradients of the FIRST minibatch (before any optimiser step) and its logged scalars
    perm = np.array([2, 0, 3, 1])
    grads = {}

    def hook(name, net, mb):
        if mb == 0:
            grads[name] = net.to_tree(net.grads)
    eng.grad_hook = hook
    info0 = {}
    Eb = bs // T_
    tg_np = {k: (v.cpu().numpy() if torch.is_tensor(v) else v) for k, v in tg.items()}
    want = R.minibatch_losses(leaf, ocfg, r, d, tg_np, perm[:Eb], hpd, eng.eps_hat.cpu())
    info = eng.update(ro, det, step, perm)
    _check_first_minibatch_grads(leaf, grads, ("Vl", "Vh", "policy"))
    # info reports the LAST minibatch, after one optimiser step: loose tolerance, keys as in the reference
    for k in ("Vl/loss", "Vl/grad_norm", "Vl/has_nan", "Vl/max_target", "Vl/min_target", "Vh/loss_Vh", "Vh/grad_Vh_norm",
              "Vh/grad_Vh_has_nan", "policy/loss", "policy/grad_norm", "policy/has_nan", "policy/log_pi_min",
              "policy/clip_frac", "policy/entropy", "policy/total_variation_dist", "eval/safe_data"):
        assert k in info and np.isfinite(info[k]), k
    assert abs(info["eval/safe_data"] - wt["safe"]) < 0.05
    assert info["Vl/has_nan"] == 0.0 and float(eng.opt["policy"].state[2]) == B // Eb


def test_rollout_hip_graph_replay_is_bit_exact(cuda):
    """Engine(use_graphs=True): eager first call, captured second call, replayed afterwards — every rollout must equal
    the eager engine's bit for bit, also after the policy parameters changed in place between calls."""
    from dgppo_amd import engine as EN
    B, T_ = 8, 6
    cfg, ocfg, hp, eng_e, trees = _setup("LidarSpread", 3, 2, B, T_, cuda, 16, 3)
    eng_g = EN.Engine(cfg, hp, cuda, T=T_, use_graphs=True)
    for k, net in eng_g.nets.items():
        net.load_tree(trees[k])
    for call in range(4):
        seeds = (torch.arange(1, B + 1, dtype=torch.int64, device=cuda) + 100 * call) * 7919
        if call == 3:   # in-place parameter update, as the optimiser does it
            for e in (eng_e, eng_g):
                e.policy.params.mul_(1.01)
        for stochastic in (True, False):
            a = eng_e.rollout(seeds, stochastic, noise_seed=3 + call).finalize()
            b = eng_g.rollout(seeds, stochastic, noise_seed=3 + call).finalize()
            for name in ("agent", "hits", "actions", "log_pis", "rnn_states", "rewards", "costs"):
                x, y = getattr(a, name), getattr(b, name)
                if x is None:
                    assert y is None
                    continue
                assert torch.equal(x, y), f"call {call} stochastic={stochastic}: {name} differs"
    assert eng_g._ro_cache[(B, True)]["graph"] is not None and eng_g._ro_cache[(B, False)]["graph"] is not None


def test_multi_stream_update_equals_single_stream(cuda):
    """Engine(multi_stream=True) runs the Vl / Vh / policy updates of a minibatch on three HIP streams: after a full
    update (several minibatches, i.e. several optimiser steps per network) the parameters must agree with the
    single-stream engine to fp32 reduction-order noise."""
    from dgppo_amd import engine as EN
    B, T_, rs, bs = 8, 8, 4, 16
    cfg, ocfg, hp, eng_a, trees = _setup("LidarSpread", 3, 2, B, T_, cuda, bs, rs)
    eng_b = EN.Engine(cfg, hp, cuda, T=T_, multi_stream=True)
    for k, net in eng_b.nets.items():
        net.load_tree(trees[k])
    eng_b.set_entropy_noise(77)
    seeds = torch.arange(1, B + 1, dtype=torch.int64, device=cuda) * 7919
    perm = np.random.default_rng(5).permutation(B)
    infos = []
    for eng in (eng_a, eng_b):
        ro = eng.rollout(seeds, True, noise_seed=3)
        det = eng.rollout(seeds + 1000, False)
        infos.append(eng.update(ro, det, 10, perm))
        torch.cuda.synchronize()
    for name in ("Vl", "Vh", "policy"):
        pa, pb = eng_a.nets[name].params, eng_b.nets[name].params
        assert float(eng_a.opt[name].state[2]) == float(eng_b.opt[name].state[2]) == B // (bs // T_)
        err = float((pa - pb).abs().max())
        assert err <= 2e-5 * max(1.0, float(pa.abs().max())), f"{name}: parameters differ by {err:.3e}"
    for k in ("Vl/loss", "Vh/loss_Vh", "policy/loss", "policy/entropy"):
        assert abs(infos[0][k] - infos[1][k]) <= 1e-4 * max(1.0, abs(infos[0][k])), k


def test_informarl_targets_and_gradients(cuda):
    """Engine(algo="informarl"): Vl pass, shaped stage cost, Dec-OCP GAE with Vh := Vl, normalised
    advantage and the first minibatch's Vl / policy gradients against the oracle; cost-weight schedule x5 past 50 %."""
    from dgppo_amd import engine as EN, init
    from oracle import algo_ref as A
    kind, n, n_obs, B, T_, rs, bs = "LidarSpread", 3, 2, 4, 8, 4, 16
    cfg, ocfg, hp0, eng0, trees = _setup(kind, n, n_obs, B, T_, cuda, bs, rs)
    hp = EN.Hyper(batch_size=bs, rnn_step=rs, train_steps=100, cost_weight=0.3, cost_schedule=True)
    eng = EN.Engine(cfg, hp, cuda, T=T_, algo="informarl")
    assert set(eng.nets) == {"policy", "Vl"}
    for k, net in eng.nets.items():
        net.load_tree(trees[k])
    eng.set_entropy_noise(77)
    seeds = torch.arange(1, B + 1, dtype=torch.int64, device=cuda) * 7919
    ro = eng.rollout(seeds, True, noise_seed=3).finalize()
    step = 60
    w = eng.cost_weight_at(step)
    assert w == pytest.approx(1.5) and A.cost_weight_schedule(0.3, step, 100, True) == pytest.approx(1.5)
    tg = eng.targets_informarl(ro, step)
    r = _np_rollout(ro)
    hpd = dict(gamma=hp.gamma, gae_lambda=hp.gae_lambda, rnn_step=rs, clip_eps=hp.clip_eps, coef_ent=hp.coef_ent)
    leaf = {k: T.tree_map(lambda t: t.clone().requires_grad_(), trees[k]) for k in ("policy", "Vl")}
    wt = R.targets_informarl(leaf, ocfg, r, hpd, w)
    _close(tg["Vl"], wt["Vl"], "Vl")
    _close(tg["Ql"], wt["Ql"], "Ql")
    # the advantage is standardised per env over T = 8 samples, which divides any input difference by std(Al):
    # (1) the kernel on the device's own Ql / Vl against the formula (informarl.py:334-336) on the same inputs: 1e-5;
    # (2) end to end: within the propagated bound 4 (dQl + dVl) / std
    gQl, gVl, gadv = (tg[k].cpu().numpy() for k in ("Ql", "Vl", "adv"))
    Al = gQl - gVl[:, :-1]
    same_in = -((Al - Al.mean(1, keepdims=True)) / (Al.std(1, keepdims=True) + 1e-8))
    _close(gadv, np.repeat(same_in[:, :, None], n, axis=-1), "advantage kernel on identical inputs")
    Al_o = wt["Ql"] - wt["Vl"][:, :-1]
    bound = 4.0 * (np.abs(gQl - wt["Ql"]).max() + np.abs(gVl - wt["Vl"]).max()) / (Al_o.std(1, keepdims=True) + 1e-8) + 1e-5
    assert (np.abs(gadv - wt["adv"]) <= bound[:, :, None]).all()
    perm = np.array([2, 0, 3, 1])
    grads = {}

    def hook(name, net, mb):
        if mb == 0:
            grads[name] = net.to_tree(net.grads)
    eng.grad_hook = hook
    Eb = bs // T_
    tg_np = {k: v.cpu().numpy() for k, v in tg.items()}
    want = R.minibatch_losses(leaf, ocfg, r, None, tg_np, perm[:Eb], hpd, eng.eps_hat.cpu())
    info = eng.update(ro, None, step, perm)
    assert set(grads) == {"Vl", "policy"} and "Vh/loss_Vh" not in info and "Vh/loss_Vh" not in want
    _check_first_minibatch_grads(leaf, grads, ("Vl", "policy"))
    for k in ("Vl/loss", "Vl/grad_norm", "policy/loss", "policy/entropy", "policy/clip_frac"):
        assert k in info and np.isfinite(info[k]), k
    assert float(eng.opt["policy"].state[2]) == B // Eb


@pytest.mark.parametrize("kind,n,n_obs", [("LidarSpread", 3, 2), ("MPESpread", 3, 3)])
def test_hcbfcrpo_targets_and_gradients(cuda, kind, n, n_obs):
    """Engine(algo="hcbfcrpo"): hand-crafted CBF Vh := get_cost(graph) incl. the final graph, Dec-OCP GAE,
    DGPPO's advantage merge, Vl / policy gradients of the first minibatch — against the oracle."""
    from dgppo_amd import engine as EN
    B, T_, rs, bs = 4, 8, 4, 16
    cfg, ocfg, hp0, eng0, trees = _setup(kind, n, n_obs, B, T_, cuda, bs, rs)
    hp = EN.Hyper(batch_size=bs, rnn_step=rs, train_steps=100)
    eng = EN.Engine(cfg, hp, cuda, T=T_, algo="hcbfcrpo")
    for k, net in eng.nets.items():
        net.load_tree(trees[k])
    eng.set_entropy_noise(77)
    seeds = torch.arange(1, B + 1, dtype=torch.int64, device=cuda) * 7919
    ro = eng.rollout(seeds, True, noise_seed=3).finalize()
    step = 80                                                       # past 75 %: schedule weight x4
    tg = eng.targets_hcbfcrpo(ro, step)
    r = _np_rollout(ro)
    hpd = dict(gamma=hp.gamma, gae_lambda=hp.gae_lambda, alpha=hp.alpha, cbf_eps=hp.cbf_eps, rnn_step=rs,
               clip_eps=hp.clip_eps, coef_ent=hp.coef_ent)
    leaf = {k: T.tree_map(lambda t: t.clone().requires_grad_(), trees[k]) for k in ("policy", "Vl")}
    assert eng.cbf_weight_at(step) == 4.0
    wt = R.targets_hcbfcrpo(leaf, ocfg, r, hpd, 4.0)
    np.testing.assert_array_equal(tg["Vh"].cpu().numpy()[:, :T_], r["costs"])            # stored costs ARE get_cost(graph)
    np.testing.assert_allclose(tg["Vh"].cpu().numpy()[:, T_], wt["Vh"][:, T_], atol=1e-6)   # cost of next_graph[-1]
    for k in ("Vl", "Ql", "Qh"):
        _close(tg[k], wt[k], k)
    _check_advantage(tg, wt, ocfg.dt, hp.alpha, hp.cbf_eps, 4.0, "hcbfcrpo " + kind)
    perm = np.array([2, 0, 3, 1])
    grads = {}

    def hook(name, net, mb):
        if mb == 0:
            grads[name] = net.to_tree(net.grads)
    eng.grad_hook = hook
    Eb = bs // T_
    tg_np = {k: v.cpu().numpy() for k, v in tg.items()}
    R.minibatch_losses(leaf, ocfg, r, None, tg_np, perm[:Eb], hpd, eng.eps_hat.cpu())
    info = eng.update(ro, None, step, perm)
    _check_first_minibatch_grads(leaf, grads, ("Vl", "policy"))
    assert "eval/safe_data" in info and "Vh/loss_Vh" not in info and abs(info["eval/safe_data"] - wt["safe"]) < 0.05


@pytest.mark.parametrize("kind,n,n_obs,B,T_,rs,blk", [
    ("LidarSpread", 8, 3, 8, 8, 4, 3),            # BASELINE config 3/4 topology: N = 81 nodes, fan-in 24
    ("LidarBicycleTarget", 16, 8, 5, 4, 2, 3),    # BASELINE config 5 topology: N = 161 nodes, state_dim 5, node_dim 8
    # the benchmark's HORIZON: T = 128, rnn_step = 16 -> 8 chunks per env (rnn_chunk_ids (8, 16),
    # dgppo.py:157-158), the [B, 129, ...] carry / value buffers, the env-major transpose of 129-step records, the
    # gae_cols_kernel<32, 4> instantiation inside the engine, 129-graph pre-pass blocks
    ("LidarSpread", 8, 3, 8, 128, 16, 3),
    ("LidarBicycleTarget", 16, 8, 3, 128, 16, 2),
])
def test_full_topology_multi_block_prepass_targets_and_gradients(cuda, kind, n, n_obs, B, T_, rs, blk):
    """The engine exactly as training / bench.py runs it (HIP-graph rollouts, three-stream update) at the benchmark
    TOPOLOGIES, with the value pre-passes forced into >= 3 env blocks with a ragged last one (`prepass_graphs`): at
    B = 4096, T = 128 the default blocking is 9 blocks of 508 envs, so the e0 > 0 slices / strided bases of
    Engine._block_feats and the reuse of the pre-pass scratch across blocks are on the path of every real run.
    Vl / Vh / Vh_det / Ql / Qh / Qh_det and the first minibatch's gradients against the oracle (dgppo.py:204-229,
    296-321; informarl.py:357-457), plus: the blocked result equals the single-block result (1e-6)."""
    bs = 2 * T_ if B % 2 == 0 else T_                                       # envs per minibatch: 2 (or 1 when B is odd)
    kw = dict(prepass_graphs=blk * (T_ + 1) + 1, use_graphs=True, multi_stream=True)
    cfg, ocfg, hp, eng, trees = _setup(kind, n, n_obs, B, T_, cuda, bs, rs, **kw)
    block = eng.prepass_graphs // (T_ + 1)
    assert block == blk and B % block != 0 and -(-B // block) >= 2, "needs a ragged multi-block split"
    seeds = torch.arange(1, B + 1, dtype=torch.int64, device=cuda) * 7919
    # two calls each: the second one replays the captured HIP graph (the path training uses)
    for _ in range(2):
        ro, det = eng.rollout_pair(seeds, seeds + 1000, noise_seed=3)
    torch.cuda.synchronize()
    assert eng._ro_cache[(B, True)]["graph"] is not None
    ro.finalize(); det.finalize()
    step = 60
    tg = eng.targets(ro, det, step)
    # (a) blocked == single block (row-local kernels: blocking must not change a value beyond tile-variant rounding)
    eng.prepass_graphs = 1 << 20
    tg1 = eng.targets(ro, det, step)
    for k in ("Vl", "Vh", "Vh_det", "Ql", "Qh", "Qh_det"):       # kernels pick tile variants by launch size: rounding-level only
        _close(tg[k], tg1[k].cpu().numpy(), f"{k}: blocked vs single-block pre-pass", tol=1e-6)
    eng.prepass_graphs = kw["prepass_graphs"]
    # (b) against the oracle
    hpd = dict(gamma=hp.gamma, gae_lambda=hp.gae_lambda, alpha=hp.alpha, cbf_eps=hp.cbf_eps, rnn_step=rs,
               clip_eps=hp.clip_eps, coef_ent=hp.coef_ent)
    r, d = _np_rollout(ro), _np_rollout(det)
    leaf = {k: T.tree_map(lambda t: t.clone().requires_grad_(), v) for k, v in trees.items()}
    w = eng.cbf_weight_at(step)
    wt = R.targets(leaf, ocfg, r, d, hpd, w)
    for k in ("Vl", "Vh", "Vh_det", "Ql", "Qh", "Qh_det"):
        _close(tg[k], wt[k], k)
    _check_advantage(tg, wt, ocfg.dt, hp.alpha, hp.cbf_eps, w, kind)
    # (c) first-minibatch gradients through the three-stream update
    perm = np.random.default_rng(2).permutation(B)
    grads = {}

    def hook(name, net, mb):
        if mb == 0:
            grads[name] = net.to_tree(net.grads)
    eng.grad_hook = hook
    Eb = bs // T_
    tg_np = {k: (v.cpu().numpy() if torch.is_tensor(v) else v) for k, v in tg.items()}
    R.minibatch_losses(leaf, ocfg, r, d, tg_np, perm[:Eb], hpd, eng.eps_hat.cpu())
    info = eng.update(ro, det, step, perm)
    torch.cuda.synchronize()
    _check_first_minibatch_grads(leaf, grads, ("Vl", "Vh", "policy"))
    assert info["Vl/has_nan"] == 0.0 and float(eng.opt["policy"].state[2]) == B // Eb


def test_nan_cost_skips_exactly_the_poisoned_minibatch(cuda):
    """A NaN cost in the deterministic rollout (a NaN LiDAR hit point reaches get_cost) must do what it does in the reference:
    the Dec-OCP targets of that env turn NaN (jnp.maximum propagates), the Vh loss and gradient of the minibatch holding
    the env are NaN, `has_any_nan_or_inf` fires and optax.apply_if_finite leaves the parameters and the Adam state alone —
    while the other minibatch, and the other two networks, train normally (dgppo.py:296-321, trainer/utils.py:109-118)."""
    B, T_, rs, bs = 4, 8, 4, 16
    cfg, ocfg, hp, eng, trees = _setup("LidarSpread", 3, 2, B, T_, cuda, bs, rs)
    seeds = torch.arange(1, B + 1, dtype=torch.int64, device=cuda) * 7919
    ro = eng.rollout(seeds, True, noise_seed=3)
    det = eng.rollout(seeds + 1000, False)
    det.cost_tm[5, 3, 1, 0] = float("nan")                    # env 3, step 5, agent 1
    snap = {}

    def hook(name, net, mb):
        if name == "Vh":
            snap[mb] = (net.params.detach().clone(), eng.opt["Vh"].state[:8].detach().clone())
    eng.grad_hook = hook
    before = {k: net.params.detach().clone() for k, net in eng.nets.items()}
    info = eng.update(ro, det, 10, np.asarray([0, 1, 2, 3]))  # minibatch 1 = envs {2, 3}
    torch.cuda.synchronize()
    Qh_det = eng.arena.get("tg.Qh_det", B, T_, cfg.n_agents, 2)
    assert torch.isnan(Qh_det[3, :6, 1, :]).all() and not torch.isnan(Qh_det[3, 6:]).any() and not torch.isnan(Qh_det[:3]).any()
    assert info["Vh/grad_Vh_has_nan"] == 1.0 and info["Vl/has_nan"] == 0.0 and info["policy/has_nan"] == 0.0
    assert torch.equal(snap[0][0], before["Vh"])               # minibatch 0 is entered with the initial parameters
    assert not torch.equal(snap[1][0], before["Vh"]), "minibatch 0 (clean) did not train Vh"
    assert torch.equal(eng.Vh.params, snap[1][0]), "the poisoned minibatch changed the Vh parameters"
    assert float(eng.opt["Vh"].state[2]) == float(snap[1][1][2]) == 1.0, "Adam count moved on a skipped step"
    assert torch.isfinite(eng.Vh.params).all() and torch.isfinite(eng.opt["Vh"].m).all()
    for k in ("policy", "Vl"):
        assert float(eng.opt[k].state[2]) == 2.0 and torch.isfinite(eng.nets[k].params).all()


def _two_ranks(mode, tmp_path):
    """tools/dist_rehearsal.py as two supervised ranks sharing this GPU over gloo (dgppo_amd/launch.py: per-rank log files —
    no pipe can fill up — and the job stops as soon as one rank fails) -> the two ranks' result files"""
    import io
    from dgppo_amd import launch
    prefix = str(tmp_path / mode)
    out, err = io.StringIO(), io.StringIO()
    rc = launch.spawn_ranks(os.path.join(ROOT, "tools", "dist_rehearsal.py"), ["--mode", mode, "--out", prefix], 2,
                            str(tmp_path / "ranklogs"), stall_seconds=300, out=out, err=err)
    assert rc == 0, err.getvalue()[-6000:]
    return [torch.load(f"{prefix}.r{r}.pt", weights_only=True) for r in range(2)]


def _twin(mode, cuda, tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import dist_rehearsal as DR
    return DR.run_mode(mode, cuda, 1, 0, None, 2, log_dir=str(tmp_path / "twin_logs"))


def _assert_params_follow(res, twin, tol, label):
    for it, (p0, p1, pt) in enumerate(zip(res[0]["params"], res[1]["params"], twin["params"])):
        for name in p0:
            assert torch.equal(p0[name], p1[name]), f"{label}: replicas diverged in {name} after iteration {it}"
            scale = float(pt[name].abs().max())
            err = float((p0[name] - pt[name]).abs().max())
            assert err <= tol * scale, f"{label}: {name} after iteration {it}: {err:.3e} from the single-process run (scale {scale:.3g})"


def test_two_rank_update_allreduce_equals_full_minibatch(cuda, tmp_path):
    """Engine(allreduce=..., world=2, rank=r) with the real sharding (rank r owns the global envs [4r, 4r+4): seeds and
    sampling noise are functions of the global env index), two ranks sharing this GPU over gloo: the ONE all-reduced flat
    buffer [g_policy | g_Vl | g_Vh | loss sums] / world must equal the single-process gradient on the union minibatch
    (mean of equal shards = global mean), the logged scalars must be global and equal the single-process
    ones, and both replicas must hold bit-identical parameters after all optimiser steps."""
    res = _two_ranks("eager", tmp_path)
    twin = _twin("eager", cuda, tmp_path)
    for name in ("policy", "Vl", "Vh"):
        assert torch.equal(res[0]["grads"][name], res[1]["grads"][name]), f"{name}: ranks disagree on the reduced gradient"
        got = res[0]["grads"][name] / 2.0
        scale = float(twin["grads"][name].abs().max())
        err = float((got - twin["grads"][name]).abs().max())
        assert err <= 2e-5 * max(scale, 1e-3), f"{name}: all-reduced mean gradient off by {err:.3e} (scale {scale:.3e})"
    _assert_params_follow(res, twin, 2e-5, "eager")
    i0, i1, it = res[0]["info"][0], res[1]["info"][0], twin["info"][0]
    for k in it:                                   # every logged value is global: identical on the ranks, equal to the twin's
        assert i0[k] == i1[k], f"{k}: rank 0 logs {i0[k]!r}, rank 1 {i1[k]!r}"
        assert abs(i0[k] - it[k]) <= 2e-5 * max(1.0, abs(it[k])), f"{k}: {i0[k]!r} vs single-process {it[k]!r}"


def test_two_rank_graph_replay_path_keeps_replicas_identical_and_follows_the_single_process_run(cuda, tmp_path):
    """The path bench.py and the algos really run under data parallelism: graph(body_pre) -> eager all-reduce ->
    graph(optimiser steps), use_graphs=True, no gradient hook, THREE iterations so that iterations 2 and 3 replay both
    graphs with fresh rollouts in the persistent buffers (stale pointers or an ordering bug between the two
    replays and the collective would show as diverging replicas or as drift from the single-process run)."""
    res = _two_ranks("graphs", tmp_path)
    twin = _twin("graphs", cuda, tmp_path)
    _assert_params_follow(res, twin, 5e-5, "graphs")
    for it in range(3):
        for k in ("Vl/loss", "Vh/loss_Vh", "policy/loss", "eval/safe_data"):
            assert res[0]["info"][it][k] == res[1]["info"][it][k]
            assert abs(res[0]["info"][it][k] - twin["info"][it][k]) <= 5e-5 * max(1.0, abs(twin["info"][it][k])), (it, k)


def test_two_rank_trainer_is_the_single_process_trainer_on_the_union_batch(cuda, tmp_path):
    """train.py's wiring (make_env / make_algo(allreduce, world, rank) / Trainer(rank, world).train()) for two iterations as two
    gloo ranks on this GPU: bit-identical parameters on the ranks, equal to the one-rank Trainer on the union batch (global
    n_env_train = 8, global minibatch = 4 envs), and only rank 0 evaluates, logs and saves."""
    res = _two_ranks("trainer", tmp_path)
    twin = _twin("trainer", cuda, tmp_path)
    _assert_params_follow(res, twin, 5e-5, "trainer")
    assert res[0]["wrote_logs"] and res[0]["saved_models"] == ["0", "1"]
    assert not res[1]["wrote_logs"] and res[1]["saved_models"] == []


def test_two_rank_informarl_lagr_all_reduces_the_multiplier_step(cuda, tmp_path):
    """informarl_lagr under data parallelism: the `.mean()` of the multiplier step (informarl_lagr.py:300-306) runs over the
    global minibatch — per-rank sums, one more small all-reduce, identical step on every rank."""
    res = _two_ranks("lagr", tmp_path)
    twin = _twin("lagr", cuda, tmp_path)
    _assert_params_follow(res, twin, 5e-5, "lagr")
    assert torch.equal(res[0]["lagr"], res[1]["lagr"])
    assert float((res[0]["lagr"] - twin["lagr"]).abs().max()) <= 1e-5 * float(twin["lagr"].abs().max())
    assert float((twin["lagr"] - 0.78).abs().max()) > 1e-6, "the multipliers did not move: the check would be vacuous"


@pytest.mark.parametrize("algo", ["dgppo", "informarl"])
def test_update_hip_graph_replay_equals_eager(cuda, algo):
    """Engine(use_graphs=True) captures ONE minibatch step (gathers, three forward/backward passes on three streams, clip +
    Adam) into a HIP graph during the first update and replays it for every later minibatch and iteration.  Over three
    iterations (the first captures, the others replay throughout, with fresh rollouts in the same persistent buffers)
    the parameters and the logged scalars must follow the eager engine's."""
    from dgppo_amd import engine as EN
    B, T_, rs, bs = 8, 8, 4, 16
    cfg, ocfg, hp, eng_e, trees = _setup("LidarSpread", 3, 2, B, T_, cuda, bs, rs, multi_stream=True, algo=algo)
    eng_g = EN.Engine(cfg, hp, cuda, T=T_, use_graphs=True, multi_stream=True, algo=algo)
    for k, net in eng_g.nets.items():
        net.load_tree(trees[k])
    eng_g.set_entropy_noise(77)
    for it in range(3):
        seeds = (torch.arange(1, B + 1, dtype=torch.int64, device=cuda) + 50 * it) * 7919
        perm = np.random.default_rng(it).permutation(B)
        infos = []
        for eng in (eng_e, eng_g):
            ro = eng.rollout(seeds, True, noise_seed=3 + it)
            det = eng.rollout(seeds + 1000, False) if algo == "dgppo" else None
            infos.append(eng.update(ro, det, it, perm))
            torch.cuda.synchronize()
        for name in eng_e.nets:
            pa, pb = eng_e.nets[name].params, eng_g.nets[name].params
            err = float((pa - pb).abs().max())
            assert err <= 1e-6 * max(1.0, float(pa.abs().max())), f"iteration {it}, {name}: graph replay drifted by {err:.3e}"
            assert float(eng_e.opt[name].state[2]) == float(eng_g.opt[name].state[2]) == (it + 1) * (B // (bs // T_))
        for k in infos[0]:
            assert abs(infos[0][k] - infos[1][k]) <= 1e-5 * max(1.0, abs(infos[0][k])), (it, k)
    assert eng_g._upd_graph.get("graph") is not None, "the minibatch step was never captured"


def test_informarl_lagr_targets_gradients_and_multipliers(cuda):
    """Engine(algo="informarl_lagr") (dgppo/algo/informarl_lagr.py:125-309): Vl and the global-info Vh
    scanned with their own carries, GAE on the clipped costs, the Lagrangian advantage, first-minibatch gradients of
    Vl / Vh / policy, and the multiplier update after the policy step — all against the oracle."""
    from dgppo_amd import engine as EN, init
    from oracle import algo_ref as A
    kind, n, n_obs, B, T_, rs = "LidarSpread", 3, 2, 4, 8, 4
    bs = B * T_                                                       # ONE minibatch: the multiplier is checked after it
    cfg, ocfg, hp0, eng0, trees = _setup(kind, n, n_obs, B, T_, cuda, bs, rs)
    hp = EN.Hyper(batch_size=bs, rnn_step=rs, train_steps=100, lagr_init=0.4, lr_lagr=0.05)
    eng = EN.Engine(cfg, hp, cuda, T=T_, algo="informarl_lagr", multi_stream=True)
    gen = torch.Generator().manual_seed(21)
    trees["Vh"] = T.tree_map(lambda t: t + 0.05 * torch.randn(t.shape, generator=gen), T.init_value(5, cfg.node_dim, 2, 1, global_info=True))
    for k, net in eng.nets.items():
        net.load_tree(trees[k])
    eng.set_entropy_noise(77)
    seeds = torch.arange(1, B + 1, dtype=torch.int64, device=cuda) * 7919
    ro = eng.rollout(seeds, True, noise_seed=3).finalize()
    r = _np_rollout(ro)
    hpd = dict(gamma=hp.gamma, gae_lambda=hp.gae_lambda, rnn_step=rs, clip_eps=hp.clip_eps, coef_ent=hp.coef_ent)
    lagr0 = eng.lagr.cpu().numpy().copy()
    assert np.all(lagr0 == np.float32(0.4))
    tg = eng.targets_lagr(ro, 0)
    leaf = {k: T.tree_map(lambda t: t.clone().requires_grad_(), v) for k, v in trees.items()}
    wt = R.targets_lagr(leaf, ocfg, r, hpd, lagr0)
    for k in ("Vl", "Vh", "Ql", "Qh"):
        _close(tg[k], wt[k], k)
    # standardised advantages: kernel on the device's own inputs to 1e-5, end to end within the propagated bound
    g = {k: tg[k].cpu().numpy() for k in ("Vl", "Vh", "Ql", "Qh", "adv", "Ah")}
    same_A, same_Ah = A.advantage_lagr(g["Ql"], g["Vl"], g["Qh"], g["Vh"], lagr0)
    _close(g["adv"], same_A, "lagr advantage kernel on identical inputs")
    _close(g["Ah"], same_Ah, "Ah kernel on identical inputs")
    stdh = (wt["Qh"] - wt["Vh"][:, :-1]).std(axis=1, keepdims=True) + 1e-8
    bound_h = 4.0 * (np.abs(g["Qh"] - wt["Qh"]).max() + np.abs(g["Vh"] - wt["Vh"]).max()) / stdh + 1e-5
    assert (np.abs(g["Ah"] - wt["Ah"]) <= bound_h).all()
    perm = np.arange(B)
    grads = {}

    def hook(name, net, mb):
        if mb == 0:
            grads[name] = net.to_tree(net.grads)
    eng.grad_hook = hook
    tg_np = {k: v.cpu().numpy() for k, v in tg.items()}
    R.minibatch_losses_lagr(leaf, ocfg, r, tg_np, perm, hpd, eng.eps_hat.cpu())
    info = eng.update(ro, None, 0, perm)
    torch.cuda.synchronize()
    _check_first_minibatch_grads(leaf, grads, ("Vl", "Vh", "policy"))
    for k in ("Vh/loss", "Vh/grad_norm", "Vh/has_nan", "Vh/max_target", "Vh/min_target", "policy/lagr_mean", "Vl/loss", "policy/loss"):
        assert k in info and np.isfinite(info[k]), k
    assert "eval/safe_data" not in info
    # multiplier update with the UPDATED policy (informarl_lagr.py:286-309): oracle evaluation of the device's new parameters
    new_pol = T.tree_map(lambda a: torch.from_numpy(np.ascontiguousarray(a)), eng.policy.to_tree())
    lp_new = R.log_pi_full_episode({"policy": new_pol}, ocfg, r, perm, eng.eps_hat.cpu())
    want = A.lagr_update(lagr0, lp_new, r["log_pis"], tg_np["Vh"][:, :T_], tg_np["Ah"], hp.gamma, hp.lr_lagr)
    got = eng.lagr.cpu().numpy()
    assert not np.array_equal(got, lagr0)
    np.testing.assert_allclose(got, want, atol=2e-6)
    assert abs(info["policy/lagr_mean"] - float(want.mean())) < 1e-5


@pytest.mark.parametrize("use_rnn,rnn_layers,use_lstm", [(False, 1, False), (True, 2, False), (True, 1, True), (True, 2, True)])
def test_rnn_options_targets_and_gradients(cuda, use_rnn, rnn_layers, use_lstm):
    """train.py --no-rnn, --rnn-layers 2 and --use-lstm through the whole engine (rollout with the packed carry, value pre-passes where
    the one-cell constraint-value net reads layer 0 of the actor's carry, GAE, first-minibatch gradients) vs the oracle."""
    kind, n, n_obs, B, T_, rs, bs = "LidarSpread", 3, 2, 4, 8, 4, 16
    cfg, ocfg, hp, eng, trees = _setup(kind, n, n_obs, B, T_, cuda, bs, rs, use_rnn=use_rnn, rnn_layers=rnn_layers,
                                       use_lstm=use_lstm, multi_stream=True)
    assert eng.HC == 64 * (rnn_layers if use_rnn else 1) * (2 if use_lstm else 1)
    seeds = torch.arange(1, B + 1, dtype=torch.int64, device=cuda) * 7919
    ro = eng.rollout(seeds, True, noise_seed=3)
    det = eng.rollout(seeds + 1000, False)
    ro.finalize(); det.finalize()
    assert ro.rnn_states.shape == (B, T_, n, eng.HC)
    if not use_rnn:
        assert float(ro.rnn_states.abs().max()) == 0.0            # no cell: the zero carry passes through
    step = 10
    tg = eng.targets(ro, det, step)
    hpd = dict(gamma=hp.gamma, gae_lambda=hp.gae_lambda, alpha=hp.alpha, cbf_eps=hp.cbf_eps, rnn_step=rs,
               clip_eps=hp.clip_eps, coef_ent=hp.coef_ent)
    r, d = _np_rollout(ro), _np_rollout(det)
    leaf = {k: T.tree_map(lambda t: t.clone().requires_grad_(), v) for k, v in trees.items()}
    w = eng.cbf_weight_at(step)
    wt = R.targets(leaf, ocfg, r, d, hpd, w)
    for k in ("Vl", "Vh", "Vh_det", "Ql", "Qh", "Qh_det"):
        _close(tg[k], wt[k], k)
    _check_advantage(tg, wt, ocfg.dt, hp.alpha, hp.cbf_eps, w, f"rnn={use_rnn} x{rnn_layers} lstm={use_lstm}")
    perm = np.array([2, 0, 3, 1])
    grads = {}

    def hook(name, net, mb):
        if mb == 0:
            grads[name] = net.to_tree(net.grads)
    eng.grad_hook = hook
    Eb = bs // T_
    tg_np = {k: (v.cpu().numpy() if torch.is_tensor(v) else v) for k, v in tg.items()}
    R.minibatch_losses(leaf, ocfg, r, d, tg_np, perm[:Eb], hpd, eng.eps_hat.cpu())
    info = eng.update(ro, det, step, perm)
    torch.cuda.synchronize()
    _check_first_minibatch_grads(leaf, grads, ("Vl", "Vh", "policy"))
    assert info["policy/has_nan"] == 0.0


def test_rccl_single_rank_communicator_on_hardware(cuda):
    """The data plane of SURVEY §8(e) on real hardware as far as a one-GPU box allows: dgppo_comm_unique_id / _init /
    _allreduce_sum_f32 / _destroy resolve librccl through dlopen, build a 1-rank communicator on this device and run the
    collective on torch's current stream (sum over one rank = identity); then Engine(allreduce=RCCL, world=1) must walk the
    data-parallel branch of update() (flat buffer, all-reduce, grad_scale, deterministic norm) to the same parameters as the
    plain engine.  Multi-rank RCCL needs a multi-GPU node (the driver's SCALE run)."""
    from dgppo_amd import dist as D, engine as EN
    comm = D.RcclComm(0, 1)
    x = torch.arange(1 << 16, dtype=torch.float32, device=cuda) * 0.5 - 7.0
    want = x.clone()
    side = torch.cuda.Stream(cuda)
    side.wait_stream(torch.cuda.current_stream(cuda))
    with torch.cuda.stream(side):                       # the collective follows the caller's stream
        comm.allreduce_sum(x)
        x.mul_(2.0)
    side.synchronize()
    assert torch.equal(x, want * 2.0)
    B, T_, rs, bs = 8, 8, 4, 16
    cfg, ocfg, hp, eng_a, trees = _setup("LidarSpread", 3, 2, B, T_, cuda, bs, rs, multi_stream=True)
    # use_graphs: with an exchange the minibatch step is TWO captured graphs with the collective issued eagerly between their
    # replays (no communication call is recorded into a graph)
    eng_b = EN.Engine(cfg, hp, cuda, T=T_, multi_stream=True, use_graphs=True, allreduce=comm.allreduce_sum, world=1)
    for k, net in eng_b.nets.items():
        net.load_tree(trees[k])
    eng_b.set_entropy_noise(77)
    for it in range(3):
        seeds = (torch.arange(1, B + 1, dtype=torch.int64, device=cuda) + 50 * it) * 7919
        perm = np.random.default_rng(it).permutation(B)
        infos = []
        for eng in (eng_a, eng_b):
            ro = eng.rollout(seeds, True, noise_seed=3 + it)
            det = eng.rollout(seeds + 1000, False)
            infos.append(eng.update(ro, det, it, perm))
            torch.cuda.synchronize()
        for name in eng_a.nets:
            pa, pb = eng_a.nets[name].params, eng_b.nets[name].params
            assert float((pa - pb).abs().max()) <= 1e-6 * max(1.0, float(pa.abs().max())), (it, name)
        for k in infos[0]:
            assert abs(infos[0][k] - infos[1][k]) <= 1e-5 * max(1.0, abs(infos[0][k])), (it, k)
    assert eng_b._upd_graph.get("graph") is not None and eng_b._upd_graph.get("graph_post") is not None
    comm.destroy()


@pytest.mark.parametrize("kind,n,n_obs,B", [("MPESpread", 3, 3, 1024), ("LidarBicycleTarget", 16, 8, 1024), ("LidarSpread", 8, 3, 4096)])
def test_baseline_configs_full_iterations_keep_their_invariants(cuda, kind, n, n_obs, B):
    """BASELINE.json configs 2, 5 (per-GPU share) and 3 at full size — T = 128, batch_size 16384, HIP-graph replay and
    three streams on, two complete DGPPO iterations — through properties that do not need the oracle: states stay inside
    their limits, the bicycle's heading stays a unit vector, LiDAR hit points lie within one sensing radius of their agent
    (or are the far-away points of missed rays),
    costs stay in their clip range, rewards are non-positive, actions lie in [-1, 1], log-probabilities / targets /
    advantages / losses are finite, the deterministic rollout differs from the stochastic one, parameters move."""
    from dgppo_amd import engine as EN, init
    T_ = 128
    cfg, ocfg, hp0, eng0, trees = _setup(kind, n, n_obs, 4, 8, cuda, 32, 4)
    hp = EN.Hyper(batch_size=16384, rnn_step=16, train_steps=1000)
    eng = EN.Engine(cfg, hp, cuda, T=T_, use_graphs=True, multi_stream=True)
    eng.policy.load_tree(init.init_policy(0, cfg.node_dim, 2, hp.actor_gnn_layers))
    eng.Vl.load_tree(init.init_value(0, cfg.node_dim, 1, hp.Vl_gnn_layers, 2))
    eng.Vh.load_tree(init.init_value(0, cfg.node_dim, cfg.n_cost, hp.Vh_gnn_layers, 3))
    eng.set_entropy_noise(1)
    p0 = {k: net.params.clone() for k, net in eng.nets.items()}
    A, vl = ocfg.area_size, ocfg.vel_limit
    for it in range(2):
        seeds = torch.arange(1, B + 1, dtype=torch.int64, device=cuda) * 7919 + it
        ro, det = eng.rollout_pair(seeds, seeds + 100000, noise_seed=it + 1)
        info = eng.update(ro, det, it, np.random.default_rng(it).permutation(B))
        assert all(np.isfinite(v) for v in info.values()), info
        for r in (ro, det):
            ag = r.agent                                                   # [B, T+1, n, sd]
            assert torch.isfinite(ag).all()
            assert float(ag[..., :2].min()) >= 0.0 and float(ag[..., :2].max()) <= A
            if ocfg.is_bicycle:
                assert float(((ag[..., 2] ** 2 + ag[..., 3] ** 2) - 1.0).abs().max()) <= 1e-5
                assert float(ag[..., 4].abs().max()) <= 0.5
            else:
                assert float(ag[..., 2:4].abs().max()) <= vl
            assert float(r.actions.abs().max()) <= 1.0
            assert float(r.rewards.max()) <= 0.0 and torch.isfinite(r.rewards).all()
            c = r.costs
            assert not torch.isinf(c).any()
            cf = c[torch.isfinite(c)]                                      # a NaN cost needs a NaN hit point (parallel ray): rare, legal
            assert float(cf.min()) >= -1.0 and (not ocfg.is_lidar or float(cf.max()) <= 1.0)
            assert int(torch.isnan(c).sum()) <= c.numel() // 1000
            if r.has_hits:
                d = (r.hits - ag[..., None, :2]).norm(dim=-1)
                d = d[torch.isfinite(d)]
                # a ray either hits within the sensing radius or misses: alpha = 1e6 puts the "hit" 5e5 away (kept unclamped
                # by the reference)
                near = d <= ocfg.comm_radius * (1 + 1e-5)
                assert bool((near | (d > 1e5)).all()) and bool(near.any())
        assert torch.isfinite(ro.log_pis).all()
        assert float((ro.actions - det.actions).abs().max()) > 1e-3
    for k, net in eng.nets.items():
        assert float((net.params - p0[k]).abs().max()) > 0 and torch.isfinite(net.params).all()
    assert eng._upd_graph.get("graph") is not None
