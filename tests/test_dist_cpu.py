"""world_size-2 `gloo` test of the data-parallel scheme on the CPU: each rank computes the oracle gradient of
the DGPPO losses on ITS shard of a minibatch, the gradient buffers are all-reduced with the same helper bench.py /
Engine use, and the result must equal the single-process gradient on the whole minibatch."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _flat_grads(trees, T):
    return {k: torch.cat([(v.grad if v.grad is not None else torch.zeros_like(v)).reshape(-1) for _, v in T.tree_leaves(tr)])
            for k, tr in trees.items()}


def _problem():
    sys.path.insert(0, ROOT)
    from oracle import dgppo_ref as R, env_np as E, nn_torch as T, train_ref
    ocfg = E.EnvCfg(E.MPE_SPREAD, n_agents=3, n_obs=2)
    trees = train_ref.init_trees(ocfg, 0)
    rng = np.random.default_rng(0)
    B, T_ = 4, 8
    ro = train_ref.rollout(ocfg, trees, rng.integers(1, 2 ** 60, size=B), T_, True, rng)
    det = train_ref.rollout(ocfg, trees, rng.integers(1, 2 ** 60, size=B), T_, False, rng)
    hp = dict(train_ref.HP, rnn_step=4)
    with torch.no_grad():
        tg = R.targets(trees, ocfg, ro, det, hp, 1.0)
    eps_hat = torch.from_numpy(rng.standard_normal((3, 2)).astype(np.float32))
    return R, T, ocfg, trees, ro, det, hp, tg, eps_hat


def _grads_on(idx):
    R, T, ocfg, trees, ro, det, hp, tg, eps_hat = _problem()
    leaf = {k: T.tree_map(lambda t: t.clone().requires_grad_(), v) for k, v in trees.items()}
    R.minibatch_losses(leaf, ocfg, ro, det, tg, np.asarray(idx), hp, eps_hat)
    return _flat_grads(leaf, T)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    torch.set_num_threads(2)
    from dgppo_amd import dist as D
    r, w = D.init_control_plane()
    assert (r, w) == (rank, world)
    ar, close = D.make_allreduce(w, backend="gloo")
    shard = [[0, 1], [2, 3]][rank]                       # equal-sized shards of the 4-env minibatch
    g = _grads_on(shard)
    # the engine's layout: ONE flat buffer [g_policy | g_Vl | g_Vh], one collective, 1/world applied by the optimiser
    keys = sorted(g)
    flat = torch.cat([g[k] for k in keys])
    ar(flat)
    flat = flat * (1.0 / w)                              # what dgppo_clip_adam_step's grad_scale does on the device
    off = 0
    for k in keys:
        g[k] = flat[off:off + g[k].numel()].clone()
        off += g[k].numel()
    t = D.max_over_ranks(float(rank + 1), w)
    assert t == float(world)
    if rank == 0:
        torch.save(g, out)
    D.barrier(w)
    close()
    D.shutdown(w)


def test_gradient_allreduce_equals_full_batch(tmp_path):
    out = str(tmp_path / "g.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out, weights_only=True)
    want = _grads_on([0, 1, 2, 3])
    for k in want:
        scale = float(want[k].abs().max())
        err = float((got[k] - want[k]).abs().max())
        assert err <= 2e-5 * max(scale, 1e-3), f"{k}: err {err:.3e} scale {scale:.3e}"


def test_shard_seeds_depend_on_global_index_only():
    sys.path.insert(0, ROOT)
    from dgppo_amd import dist as D
    whole = D.shard_seeds(0, 8, 3)
    halves = np.concatenate([D.shard_seeds(0, 4, 3), D.shard_seeds(1, 4, 3)])
    np.testing.assert_array_equal(whole, halves)
    assert len(set(whole.tolist())) == 8 and not np.array_equal(whole, D.shard_seeds(0, 8, 4))
    assert D.make_allreduce(1)[0] is None
