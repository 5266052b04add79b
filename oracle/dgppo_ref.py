"""
ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product path (dgppo_amd/).

Torch-CPU restatement of DGPPO.update_inner for ONE minibatch (dgppo/algo/dgppo.py:188-294, informarl.py:357-457,
dgppo.py:296-321) on top of oracle/nn_torch.py (per-edge networks), oracle/env_np.py (graphs) and oracle/algo_ref.py (GAE).
PARITY UNPINNED w.r.t. the reference's own outputs; it pins the HIP engine's orchestration: carry conventions
(SURVEY A.8), chunking with zero initial carry, minibatch indexing, loss definitions and gradients.
"""
from __future__ import annotations

import numpy as np
import torch

from . import algo_ref as A
from . import env_np as E
from . import nn_torch as T


def graphs_of(ocfg, agent, goal, obst, hits):
    """agent [B,T1,n,sd], hits [B,T1,n,k,2] -> torch graph dict with leading axes [B,T1]."""
    B, T1 = agent.shape[:2]
    flat = lambda x: None if x is None else x.reshape((B * T1,) + x.shape[2:])
    rep = lambda x: None if x is None else np.repeat(x, T1, axis=0)
    g = E.get_graph(ocfg, flat(agent), rep(goal), rep(obst), flat(hits))
    return {k: torch.from_numpy(v).view((B, T1) + v.shape[1:]) for k, v in g.items()}


def _sel(g, b=None, t=None):
    out = {}
    for k, v in g.items():
        x = v
        if b is not None:
            x = x[b]
        if t is not None:
            x = x[:, t]
        out[k] = x
    return out


def values_Vl(trees, ocfg, ro):
    """Vl [B,T+1]: scan_Vl over the T graphs + the final value on next_graph[-1] (informarl.py:281-320)."""
    n = ocfg.n_agents
    g = graphs_of(ocfg, ro["agent"], ro["goal"], ro["obst"], ro["hits"])
    B, T1 = ro["agent"].shape[:2]
    with torch.no_grad():
        h = torch.zeros(B, 1, T.carry_width(trees["Vl"]))
        vs = []
        for t in range(T1):
            v, h = T.value_Vl(trees["Vl"], _sel(g, t=t), h, n)
            vs.append(v)
    return torch.stack(vs, 1).numpy()


def targets_informarl(trees, ocfg, ro, hp, cost_weight):
    Vl = values_Vl(trees, ocfg, ro)
    Ql, adv = A.informarl_targets(ro["costs"], ro["rewards"], Vl, hp["gamma"], hp["gae_lambda"], cost_weight)
    return dict(Vl=Vl, Ql=Ql, adv=adv)


def targets_hcbfcrpo(trees, ocfg, ro, hp, cbf_weight):
    """hcbfcrpo.py:120-186: Vh := env.get_cost(graph) on the T rollout graphs and on next_graph[-1]."""
    Vl = values_Vl(trees, ocfg, ro)
    Tn = ro["agent"].shape[1] - 1
    hits_fin = None if ro["hits"] is None else ro["hits"][:, Tn]
    fin = E.env_step(ocfg, ro["agent"][:, Tn], ro["goal"], ro["obst"], hits_fin,
                     np.zeros((ro["agent"].shape[0], ocfg.n_agents, 2), np.float32), E.ray_table(ocfg.n_rays))
    Vh = np.concatenate([ro["costs"], fin["cost"][:, None]], axis=1).astype(np.float32)
    Qh, Ql = A.gae_batch(ro["costs"], ro["rewards"], Vh, Vl, hp["gamma"], hp["gae_lambda"])
    adv, safe = A.advantage(Ql, Vl, Vh, ocfg.dt, hp["alpha"], hp["cbf_eps"], cbf_weight)
    return dict(Vl=Vl, Vh=Vh, Ql=Ql, Qh=Qh, adv=adv, safe=safe)


def values(trees, ocfg, ro, stochastic):
    """-> Vl [B,T+1] (stochastic only), Vh [B,T+1,n,nh].  ro: dict of numpy arrays (env-major)."""
    n = ocfg.n_agents
    g = graphs_of(ocfg, ro["agent"], ro["goal"], ro["obst"], ro["hits"])
    B, T1 = ro["agent"].shape[:2]
    Tn = T1 - 1
    rnn = torch.from_numpy(ro["rnn_states"])                     # [B,T,n,L*64] stored (packed) carry
    with torch.no_grad():
        Vl = None
        if stochastic:
            h = torch.zeros(B, 1, T.carry_width(trees["Vl"]))
            vs = []
            for t in range(T1):                                  # scan_Vl + final value on next_graph[-1] (dgppo.py:204-216)
                v, h = T.value_Vl(trees["Vl"], _sel(g, t=t), h, n)
                vs.append(v)
            Vl = torch.stack(vs, 1).numpy()
        flat = {k: v[:, :Tn].reshape((B * Tn,) + v.shape[2:]) for k, v in g.items()}
        HC = rnn.shape[-1]                                       # packed actor carry; the one-cell Vh reads layer 0 (rnn.py:20)
        Vh, _ = T.value_Vh(trees["Vh"], flat, rnn.reshape(B * Tn, n, HC), n)          # dgppo.py:219-220
        Vh = Vh.view(B, Tn, n, -1)
        _, hstar = T.policy_net(trees["policy"], _sel(g, t=Tn), rnn[:, -1], n)        # dgppo.py:222-226
        Vh_fin, _ = T.value_Vh(trees["Vh"], _sel(g, t=Tn), hstar, n)
        Vh = torch.cat([Vh, Vh_fin[:, None]], 1).numpy()
    return Vl, Vh


def targets(trees, ocfg, ro, det, hp, cbf_weight):
    Vl, Vh = values(trees, ocfg, ro, True)
    _, Vh_det = values(trees, ocfg, det, False)
    Qh, Ql = A.gae_batch(ro["costs"], ro["rewards"], Vh, Vl, hp["gamma"], hp["gae_lambda"])
    Qh_det, _ = A.gae_batch(det["costs"], det["rewards"], Vh_det, Vl, hp["gamma"], hp["gae_lambda"])
    adv, safe = A.advantage(Ql, Vl, Vh, ocfg.dt, hp["alpha"], hp["cbf_eps"], cbf_weight)
    return dict(Vl=Vl, Vh=Vh, Vh_det=Vh_det, Ql=Ql, Qh=Qh, Qh_det=Qh_det, adv=adv, safe=safe)


def minibatch_losses(trees, ocfg, ro, det, tg, idx, hp, eps_hat):
    """losses + autograd gradients of the three networks for the minibatch `idx` (env indices).
    trees must have requires_grad leaves."""
    n = ocfg.n_agents
    rs = hp["rnn_step"]
    B, T1 = ro["agent"].shape[:2]
    Tn = T1 - 1
    C = Tn // rs
    sub = lambda d, k: None if d[k] is None else d[k][idx]
    g = graphs_of(ocfg, ro["agent"][idx][:, :Tn], sub(ro, "goal"), sub(ro, "obst"),
                  None if ro["hits"] is None else ro["hits"][idx][:, :Tn])
    Eb = len(idx)
    chunk = lambda v: v.reshape((Eb * C, rs) + v.shape[2:])       # (env, chunk) groups, zero initial carry
    gc = {k: chunk(v) for k, v in g.items()}
    out = {}
    # ---- Vl
    h = torch.zeros(Eb * C, 1, T.carry_width(trees["Vl"]))
    vs = []
    for tau in range(rs):
        v, h = T.value_Vl(trees["Vl"], _sel(gc, t=tau), h, n)
        vs.append(v)
    v_pred = torch.stack(vs, 1).reshape(Eb, Tn)
    loss_Vl = (0.5 * (v_pred - torch.from_numpy(tg["Ql"][idx])) ** 2).mean()
    loss_Vl.backward()
    out["Vl/loss"] = float(loss_Vl.detach())
    # ---- Vh on the deterministic rollout with ITS stored carry (DGPPO only: det is None for InforMARL)
    if det is not None:
        gd = graphs_of(ocfg, det["agent"][idx][:, :Tn], sub(det, "goal"), sub(det, "obst"),
                       None if det["hits"] is None else det["hits"][idx][:, :Tn])
        flat = {k: v.reshape((Eb * Tn,) + v.shape[2:]) for k, v in gd.items()}
        vh, _ = T.value_Vh(trees["Vh"], flat, torch.from_numpy(det["rnn_states"][idx]).reshape(Eb * Tn, n, -1), n)
        loss_Vh = (0.5 * (vh.view(Eb, Tn, n, -1) - torch.from_numpy(tg["Qh_det"][idx])) ** 2).mean()
        loss_Vh.backward()
        out["Vh/loss_Vh"] = float(loss_Vh.detach())
    # ---- policy
    a_in = chunk(torch.from_numpy(ro["actions"][idx]))
    h = torch.zeros(Eb * C, n, T.carry_width(trees["policy"]))
    lps, ents = [], []
    for tau in range(rs):
        lp, ent, h = T.policy_eval(trees["policy"], _sel(gc, t=tau), a_in[:, tau], h, n, eps_hat)
        lps.append(lp); ents.append(ent)
    lp = torch.stack(lps, 1).reshape(Eb, Tn, n)
    ent = torch.stack(ents, 1).reshape(Eb, Tn, n)
    rho = torch.exp(lp - torch.from_numpy(ro["log_pis"][idx]))
    Aadv = torch.from_numpy(tg["adv"][idx])
    l1 = -rho * Aadv
    l2 = -torch.clamp(rho, 1 - hp["clip_eps"], 1 + hp["clip_eps"]) * Aadv
    loss_pol = torch.maximum(l1, l2).mean() - hp["coef_ent"] * ent.mean()
    loss_pol.backward()
    out.update({"policy/loss": float(loss_pol.detach()), "policy/clip_frac": float((l2 > l1).float().mean()),
                "policy/entropy": float(ent.mean().detach()), "policy/total_variation_dist": float(0.5 * (rho - 1).abs().mean().detach())})
    return out


# ----------------------------------------------------------------------------------------------------------------------
# InforMARL-Lagrangian (dgppo/algo/informarl_lagr.py:125-309)
# ----------------------------------------------------------------------------------------------------------------------
def values_Vh_lagr(trees, ocfg, ro):
    """scan_Vh over the T graphs with the net's OWN zero-initialised carry + the final value on next_graph[-1]
    (informarl_lagr.py:151-161,194-207): Vh [B,T+1,n,nh].  The net is DecRStateFn(use_global_info=True)."""
    n = ocfg.n_agents
    g = graphs_of(ocfg, ro["agent"], ro["goal"], ro["obst"], ro["hits"])
    B, T1 = ro["agent"].shape[:2]
    with torch.no_grad():
        h = torch.zeros(B, n, T.carry_width(trees["Vh"]))
        vs = []
        for t in range(T1):
            v, h = T.value_Vh(trees["Vh"], _sel(g, t=t), h, n, global_info=True)
            vs.append(v)
    return torch.stack(vs, 1).numpy()


def targets_lagr(trees, ocfg, ro, hp, lagr):
    Vl = values_Vl(trees, ocfg, ro)
    Vh = values_Vh_lagr(trees, ocfg, ro)
    Qh, Ql = A.gae_batch(np.maximum(ro["costs"], 0.0), ro["rewards"], Vh, Vl, hp["gamma"], hp["gae_lambda"])
    adv, Ah = A.advantage_lagr(Ql, Vl, Qh, Vh, lagr)
    return dict(Vl=Vl, Vh=Vh, Ql=Ql, Qh=Qh, adv=adv, Ah=Ah)


def minibatch_losses_lagr(trees, ocfg, ro, tg, idx, hp, eps_hat):
    """Vl / policy losses as InforMARL, plus update_Vh of the Lagrangian baseline: chunks of rnn_step with zero initial
    carry against Qh (informarl_lagr.py:252-284).  trees must have requires_grad leaves."""
    out = minibatch_losses({"policy": trees["policy"], "Vl": trees["Vl"]}, ocfg, ro, None, tg, idx, hp, eps_hat)
    n, rs = ocfg.n_agents, hp["rnn_step"]
    B, T1 = ro["agent"].shape[:2]
    Tn = T1 - 1
    C = Tn // rs
    sub = lambda d, k: None if d[k] is None else d[k][idx]
    g = graphs_of(ocfg, ro["agent"][idx][:, :Tn], sub(ro, "goal"), sub(ro, "obst"),
                  None if ro["hits"] is None else ro["hits"][idx][:, :Tn])
    Eb = len(idx)
    gc = {k: v.reshape((Eb * C, rs) + v.shape[2:]) for k, v in g.items()}
    h = torch.zeros(Eb * C, n, T.carry_width(trees["Vh"]))
    vs = []
    for tau in range(rs):
        v, h = T.value_Vh(trees["Vh"], _sel(gc, t=tau), h, n, global_info=True)
        vs.append(v)
    vh = torch.stack(vs, 1).reshape(Eb, Tn, n, -1)
    loss_Vh = (0.5 * (vh - torch.from_numpy(tg["Qh"][idx])) ** 2).mean()
    loss_Vh.backward()
    out["Vh/loss"] = float(loss_Vh.detach())
    return out


def log_pi_full_episode(trees, ocfg, ro, idx, eps_hat):
    """log pi of the stored actions re-evaluated over the WHOLE episode from a zero carry (update_lagr,
    informarl_lagr.py:287-299): [len(idx), T, n]."""
    n = ocfg.n_agents
    B, T1 = ro["agent"].shape[:2]
    Tn = T1 - 1
    sub = lambda d, k: None if d[k] is None else d[k][idx]
    g = graphs_of(ocfg, ro["agent"][idx][:, :Tn], sub(ro, "goal"), sub(ro, "obst"),
                  None if ro["hits"] is None else ro["hits"][idx][:, :Tn])
    acts = torch.from_numpy(ro["actions"][idx])
    h = torch.zeros(len(idx), n, T.carry_width(trees["policy"]))
    lps = []
    with torch.no_grad():
        for t in range(Tn):
            lp, _, h = T.policy_eval(trees["policy"], _sel(g, t=t), acts[:, t], h, n, eps_hat)
            lps.append(lp)
    return torch.stack(lps, 1).numpy()
