"""End-to-end GPU parity of the DGPPO engine (rollout -> value pre-passes -> GAE -> advantage -> minibatch gradients)
against the torch/numpy oracle, on small configurations; plus rollout-level checks."""
import numpy as np
import pytest
import torch

from oracle import dgppo_ref as R
from oracle import env_np as E
from oracle import nn_torch as T

pytestmark = pytest.mark.gpu


def _np_rollout(ro):
    c = lambda x: None if x is None else x.detach().cpu().numpy()
    return dict(agent=c(ro.agent), hits=c(ro.hits), goal=c(ro.goal), obst=c(ro.obst), actions=c(ro.actions),
                log_pis=c(ro.log_pis), rnn_states=c(ro.rnn_states.contiguous()), rewards=c(ro.rewards), costs=c(ro.costs))


def _setup(kind_name, n, n_obs, B, T_, cuda, batch_size, rnn_step):
    from dgppo_amd import _native as N, engine as EN, init
    kind = N.ENV_KINDS[kind_name]
    cfg = N.make_env_cfg(kind, n, n_obs)
    ocfg = E.EnvCfg(kind, n_agents=n, n_obs=n_obs)
    hp = EN.Hyper(batch_size=batch_size, rnn_step=rnn_step, train_steps=100)
    eng = EN.Engine(cfg, hp, cuda, T=T_)
    trees = {"policy": init.init_policy(0, cfg.node_dim, 2, 2), "Vl": init.init_value(0, cfg.node_dim, 1, 2, 2),
             "Vh": init.init_value(0, cfg.node_dim, 2, 1, 3)}
    rng = np.random.default_rng(11)
    jitter = lambda tr: T.tree_map(lambda a: torch.from_numpy(a + 0.05 * rng.standard_normal(a.shape).astype(np.float32)), tr)
    trees = {k: jitter(v) for k, v in trees.items()}
    trees["policy"]["params"]["ScaleHid"]["kernel"] = T.orthogonal(torch.Generator().manual_seed(1), 64, 64, 0.5)
    for k, net in eng.nets.items():
        net.load_tree(trees[k])
    eng.set_entropy_noise(77)
    return cfg, ocfg, hp, eng, trees


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
                    np.testing.assert_allclose(r["log_pis"][:, t], lp.numpy(), atol=2e-5)
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


@pytest.mark.parametrize("kind,n,n_obs", [("LidarSpread", 3, 2), ("MPESpread", 3, 3)])
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
        np.testing.assert_allclose(tg[k].cpu().numpy(), wt[k], atol=2e-5, err_msg=k)
    got_adv = tg["adv"].cpu().numpy()
    assert (np.abs(got_adv - wt["adv"]) > 1e-4 * np.maximum(1, np.abs(wt["adv"]))).mean() < 0.02   # hard safe-gate flips
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
    for name in ("Vl", "Vh", "policy"):
        w = dict(T.tree_leaves(T.tree_map(lambda t: t.grad if t.grad is not None else torch.zeros_like(t), leaf[name])))
        gt = dict(T.tree_leaves(T.tree_map(lambda a: torch.from_numpy(np.ascontiguousarray(a)), grads[name])))
        scale = max(float(v.abs().max()) for v in w.values())
        for k in w:
            err = float((gt[k].double() - w[k].double()).abs().max())
            assert err <= 5e-5 * max(scale, 1e-3), f"{name} grad {k}: err {err:.3e} scale {scale:.3e}"
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
    np.testing.assert_allclose(tg["Vl"].cpu().numpy(), wt["Vl"], atol=2e-5)
    np.testing.assert_allclose(tg["Ql"].cpu().numpy(), wt["Ql"], atol=5e-5)
    np.testing.assert_allclose(tg["adv"].cpu().numpy(), wt["adv"], atol=2e-3)       # standardised: error / std of 8 samples
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
    for name in ("Vl", "policy"):
        wg = dict(T.tree_leaves(T.tree_map(lambda t: t.grad if t.grad is not None else torch.zeros_like(t), leaf[name])))
        gt = dict(T.tree_leaves(T.tree_map(lambda a: torch.from_numpy(np.ascontiguousarray(a)), grads[name])))
        scale = max(float(v.abs().max()) for v in wg.values())
        for k in wg:
            err = float((gt[k].double() - wg[k].double()).abs().max())
            assert err <= 5e-5 * max(scale, 1e-3), f"{name} grad {k}: err {err:.3e} scale {scale:.3e}"
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
        np.testing.assert_allclose(tg[k].cpu().numpy(), wt[k], atol=5e-5, err_msg=k)
    got_adv = tg["adv"].cpu().numpy()
    assert (np.abs(got_adv - wt["adv"]) > 2e-3 * np.maximum(1, np.abs(wt["adv"]))).mean() < 0.02      # hard safe-gate flips
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
    for name in ("Vl", "policy"):
        wg = dict(T.tree_leaves(T.tree_map(lambda t: t.grad if t.grad is not None else torch.zeros_like(t), leaf[name])))
        gt = dict(T.tree_leaves(T.tree_map(lambda a: torch.from_numpy(np.ascontiguousarray(a)), grads[name])))
        scale = max(float(v.abs().max()) for v in wg.values())
        for k in wg:
            err = float((gt[k].double() - wg[k].double()).abs().max())
            assert err <= 5e-5 * max(scale, 1e-3), f"{name} grad {k}: err {err:.3e} scale {scale:.3e}"
    assert "eval/safe_data" in info and "Vh/loss_Vh" not in info and abs(info["eval/safe_data"] - wt["safe"]) < 0.05
