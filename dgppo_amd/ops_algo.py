"""Torch-tensor wrappers over the GAE / advantage / optimiser entry points of the C ABI (include/dgppo_hip.h)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _native as N


def lam_pow_table(gae_lambda: float, T: int, device) -> torch.Tensor:
    return torch.from_numpy((np.float64(gae_lambda) ** np.arange(T + 1)).astype(np.float32)).to(device)


def gae(costs, rewards, Vh, Vl, lam_pow, gamma: float, gae_lambda: float, Qh, Ql):
    B, T, n, nh = costs.shape
    N.expect_shape(rewards, (B, T), "rewards")
    N.expect_shape(Vh, (B, T + 1, n, nh), "Vh")
    N.expect_shape(Vl, (B, T + 1), "Vl")
    N.expect_shape(lam_pow, (T + 1,), "lam_pow")
    N.expect_shape(Qh, (B, T, n, nh), "Qh")
    N.expect_shape(Ql, (B, T), "Ql")
    rc = N.lib().dgppo_gae(N.ptr(costs), N.ptr(rewards), N.ptr(Vh), N.ptr(Vl), N.ptr(lam_pow), C.c_float(gamma),
                           C.c_float(1 - gamma), C.c_float(1 - gae_lambda), N.ptr(Qh), N.ptr(Ql), B, T, n, nh,
                           N.stream_ptr())
    N.check(rc, "dgppo_gae")


def shaped_reward(reward, cost, cost_weight: float, out):
    """out = reward - w * sum_{agents, components} max(cost, 0)   (InforMARL's stage cost, informarl.py:329)"""
    B, T, n, nh = cost.shape
    N.expect_shape(reward, (B, T), "reward")
    N.expect_shape(out, (B, T), "out")
    rc = N.lib().dgppo_shaped_reward(N.ptr(reward), N.ptr(cost), C.c_float(cost_weight), N.ptr(out), C.c_int64(B * T), n, nh,
                                     N.stream_ptr())
    N.check(rc, "dgppo_shaped_reward")


def advantage(Ql, Vl, Vh, dt, alpha, cbf_eps, cbf_weight, adv, stats):
    """Vh None: InforMARL's plain normalised advantage (no CBF terms)."""
    if Vh is None:
        B, T, n = adv.shape
        nh = 1
    else:
        B, T1, n, nh = Vh.shape
        T = T1 - 1
    N.expect_shape(Ql, (B, T), "Ql")
    N.expect_shape(Vl, (B, T + 1), "Vl")
    N.expect_shape(adv, (B, T, n), "adv")
    rc = N.lib().dgppo_advantage(N.ptr(Ql), N.ptr(Vl), N.ptr(Vh), C.c_float(dt), C.c_float(alpha), C.c_float(cbf_eps),
                                 C.c_float(cbf_weight), N.ptr(adv), N.ptr(stats), B, T, n, nh, N.stream_ptr())
    N.check(rc, "dgppo_advantage")


def clip_adam_step(params, grads, m, v, state, lr, max_norm, b1=0.9, b2=0.999, eps=1e-8, grad_scale=1.0):
    n = params.numel()
    for t, nm in ((grads, "grads"), (m, "m"), (v, "v")):
        N.expect_shape(t, (n,), nm)
    N.expect_shape(state, (N.OPT_STATE_FLOATS,), "state")
    rc = N.lib().dgppo_clip_adam_step(N.ptr(params), N.ptr(grads), N.ptr(m), N.ptr(v), C.c_int64(n), N.ptr(state),
                                      C.c_float(lr), C.c_float(b1), C.c_float(b2), C.c_float(eps), C.c_float(max_norm),
                                      C.c_float(grad_scale), N.stream_ptr())
    N.check(rc, "dgppo_clip_adam_step")


def advantage_lagr(Ql, Vl, Qh, Vh, lagr, adv, Ah):
    """InforMARL-Lagrangian advantage (informarl_lagr.py:219-235)"""
    B, T, n, nh = Qh.shape
    N.expect_shape(Ql, (B, T), "Ql"); N.expect_shape(Vl, (B, T + 1), "Vl"); N.expect_shape(Vh, (B, T + 1, n, nh), "Vh")
    N.expect_shape(lagr, (n, nh), "lagr"); N.expect_shape(adv, (B, T, n), "adv"); N.expect_shape(Ah, (B, T, n, nh), "Ah")
    rc = N.lib().dgppo_advantage_lagr(N.ptr(Ql), N.ptr(Vl), N.ptr(Qh), N.ptr(Vh), N.ptr(lagr), N.ptr(adv), N.ptr(Ah), B, T, n, nh,
                                      N.stream_ptr())
    N.check(rc, "dgppo_advantage_lagr")


def lagr_update(lp_new, lp_old, Vh_mb, Ah_mb, lagr, sums, gamma: float, lr: float):
    """update_lagr (informarl_lagr.py:286-309).  lp_* [Eb,T,n]; Vh_mb [Eb,T+1,n,nh] (first T steps used); Ah_mb [Eb,T,n,nh]."""
    Eb, T, n, nh = Ah_mb.shape
    N.expect_shape(lp_new, (Eb, T, n), "lp_new"); N.expect_shape(lp_old, (Eb, T, n), "lp_old")
    N.expect_shape(Vh_mb, (Eb, T + 1, n, nh), "Vh"); N.expect_shape(lagr, (n, nh), "lagr"); N.expect_shape(sums, (n * nh,), "sums")
    rc = N.lib().dgppo_lagr_update(N.ptr(lp_new), N.ptr(lp_old), N.ptr(Vh_mb), C.c_int64((T + 1) * n * nh), N.ptr(Ah_mb),
                                   N.ptr(lagr), N.ptr(sums), Eb, T, n, nh, C.c_float(1.0 - gamma), C.c_float(lr), N.stream_ptr())
    N.check(rc, "dgppo_lagr_update")


def lagr_sums(lp_new, lp_old, Vh_mb, Ah_mb, sums, gamma: float):
    """first half of update_lagr for the data-parallel path: this rank's share of the sums (+=)"""
    Eb, T, n, nh = Ah_mb.shape
    N.expect_shape(lp_new, (Eb, T, n), "lp_new"); N.expect_shape(lp_old, (Eb, T, n), "lp_old")
    N.expect_shape(Vh_mb, (Eb, T + 1, n, nh), "Vh"); N.expect_shape(sums, (n * nh,), "sums")
    rc = N.lib().dgppo_lagr_sums(N.ptr(lp_new), N.ptr(lp_old), N.ptr(Vh_mb), C.c_int64((T + 1) * n * nh), N.ptr(Ah_mb),
                                 N.ptr(sums), Eb, T, n, nh, C.c_float(1.0 - gamma), N.stream_ptr())
    N.check(rc, "dgppo_lagr_sums")


def lagr_apply(lagr, sums, rows_total: int, lr: float):
    """second half: lagr = relu(lagr + lr * sums / rows_total); sums are zeroed.  rows_total = GLOBAL n_env * T."""
    N.expect_shape(sums, (lagr.numel(),), "sums")
    rc = N.lib().dgppo_lagr_apply(N.ptr(lagr), N.ptr(sums), lagr.numel(), C.c_int64(rows_total), C.c_float(lr), N.stream_ptr())
    N.check(rc, "dgppo_lagr_apply")


def relu_fwd(x, out):
    rc = N.lib().dgppo_relu_fwd(N.ptr(x), N.ptr(out), C.c_int64(x.numel()), N.stream_ptr())
    N.check(rc, "dgppo_relu_fwd")
