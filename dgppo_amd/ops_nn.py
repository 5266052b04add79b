"""Torch-tensor wrappers over the network entry points of the C ABI (include/dgppo_hip.h).
Matrices may be row-strided views (unit inner stride): (pointer, leading dimension) are passed to the kernels."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _native as N

# Matrix-core work actually enqueued, in flops (2 per multiply-add of the UNPADDED operand shapes), accumulated by the
# wrappers below; bench.py reads it to report MFMA utilisation from executed work rather than from SURVEY's per-node
# estimate.  (Host-side bookkeeping only; the engine re-adds a rollout's count when it replays a captured HIP graph.)
FLOPS = [0.0]


def _mat(t: torch.Tensor, name: str):
    """2-D fp32 CUDA tensor with unit inner stride -> (ptr, ld, rows, cols)."""
    if t.dim() != 2:
        raise ValueError(f"{name} must be 2-D, got {tuple(t.shape)}")
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU (the HIP path has no CPU fallback)")
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32")
    if t.shape[1] > 1 and t.stride(1) != 1:
        raise ValueError(f"{name} must have unit inner stride")
    ld = t.stride(0) if t.shape[0] > 1 else max(t.shape[1], t.stride(0))
    return C.c_void_p(t.data_ptr()), int(ld), int(t.shape[0]), int(t.shape[1])


def _p(t: Optional[torch.Tensor], name="tensor", dtype=torch.float32):
    return N.ptr(t, dtype, name)


def dense_fwd(X, W, bias, Y, act: int = 0, accumulate: bool = False, trans_w: bool = False, relu_mask=None):
    """Y = act(X @ W + bias)   (trans_w: X @ W.T with W stored [N, K]); relu_mask: Y = where(relu_mask > 0, Y, 0) last."""
    xp, ldx, M, K = _mat(X, "X")
    wp, ldw, wr, wc = _mat(W, "W")
    yp, ldy, My, Ncol = _mat(Y, "Y")
    Kw, Nw = (wc, wr) if trans_w else (wr, wc)
    if Kw != K or Nw != Ncol or My != M:
        raise ValueError(f"dense_fwd: shape mismatch X{tuple(X.shape)} W{tuple(W.shape)} trans={trans_w} Y{tuple(Y.shape)}")
    if bias is not None:
        N.expect_shape(bias, (Ncol,), "bias")
    mp, ldm = C.c_void_p(0), 0
    if relu_mask is not None:
        mp, ldm, Mm, Nm = _mat(relu_mask, "relu_mask")
        if (Mm, Nm) != (M, Ncol):
            raise ValueError(f"dense_fwd: relu_mask {tuple(relu_mask.shape)} does not match Y {tuple(Y.shape)}")
    FLOPS[0] += 2.0 * M * K * Ncol
    rc = N.lib().dgppo_dense_fwd(xp, ldx, wp, ldw, _p(bias, "bias"), yp, ldy, M, K, Ncol, int(act), int(accumulate),
                                 int(trans_w), mp, ldm, N.stream_ptr())
    N.check(rc, "dgppo_dense_fwd")


def mlp_gi_fwd(X, W1, b1, g1, be1, W2, b2, g2, be2, Wi, bi, gi, saves=None):
    """gi = relu(LN(relu(LN(X W1 + b1)) W2 + b2)) Wi + bi in one kernel; saves = (p1, y1, st1, p2, y2, st2) or None."""
    xp, ldx, M, K = _mat(X, "X")
    if K != 64:
        raise ValueError(f"mlp_gi_fwd: X must be [M, 64], got {tuple(X.shape)}")
    for t, shp, nm in ((W1, (64, 64), "W1"), (W2, (64, 64), "W2"), (Wi, (64, 192), "Wi"), (b1, (64,), "b1"), (g1, (64,), "g1"),
                       (be1, (64,), "be1"), (b2, (64,), "b2"), (g2, (64,), "g2"), (be2, (64,), "be2"), (bi, (192,), "bi"),
                       (gi, (M, 192), "gi")):
        N.expect_shape(t, shp, nm)
    sv = [None] * 6
    if saves is not None:
        for t, shp, nm in zip(saves, ((M, 64), (M, 64), (M, 2), (M, 64), (M, 64), (M, 2)), ("p1", "y1", "st1", "p2", "y2", "st2")):
            N.expect_shape(t, shp, nm)
        sv = list(saves)
    FLOPS[0] += 2.0 * M * (64 * 64 * 2 + 64 * 192)
    rc = N.lib().dgppo_mlp_gi_fwd(xp, ldx, _p(W1), _p(b1), _p(g1), _p(be1), _p(W2), _p(b2), _p(g2), _p(be2), _p(Wi), _p(bi),
                                  *[_p(t) for t in sv], _p(gi), M, N.stream_ptr())
    N.check(rc, "dgppo_mlp_gi_fwd")


def mlp_gi_bwd(dgi, Wi, W2, W1, g2, g1, p2, y2, st2, p1, y1, st1, relu_mask, dpre2, dpre1, dx, dg2, db2, dg1, db1):
    """activation gradients of the mlp_gi_fwd chain in one kernel (dgppo_mlp_gi_bwd): dgi [M,192] -> dpre2, dpre1, dx [M,64];
    dg* / db* [64] accumulate the LayerNorm parameter gradients; relu_mask [M,64] or None masks dx."""
    M = dgi.shape[0]
    N.expect_shape(dgi, (M, 192), "dgi")
    for t, shp, nm in ((Wi, (64, 192), "Wi"), (W2, (64, 64), "W2"), (W1, (64, 64), "W1"), (g2, (64,), "g2"), (g1, (64,), "g1"),
                       (p2, (M, 64), "p2"), (y2, (M, 64), "y2"), (st2, (M, 2), "st2"), (p1, (M, 64), "p1"), (y1, (M, 64), "y1"),
                       (st1, (M, 2), "st1"), (dpre2, (M, 64), "dpre2"), (dpre1, (M, 64), "dpre1"), (dx, (M, 64), "dx"),
                       (dg2, (64,), "dg2"), (db2, (64,), "db2"), (dg1, (64,), "dg1"), (db1, (64,), "db1")):
        N.expect_shape(t, shp, nm)
    mp, ldm = (None, 0)
    if relu_mask is not None:
        mp, ldm, Mm, Km = _mat(relu_mask, "relu_mask")
        if (Mm, Km) != (M, 64):
            raise ValueError(f"mlp_gi_bwd: relu_mask must be [M, 64], got {tuple(relu_mask.shape)}")
    FLOPS[0] += 2.0 * M * (64 * 64 * 2 + 64 * 192)
    rc = N.lib().dgppo_mlp_gi_bwd(_p(dgi), _p(Wi), _p(W2), _p(W1), _p(g2), _p(g1), _p(p2), _p(y2), _p(st2), _p(p1), _p(y1), _p(st1),
                                  mp if mp is not None else C.c_void_p(0), ldm, _p(dpre2), _p(dpre1), _p(dx), 64,
                                  _p(dg2), _p(db2), _p(dg1), _p(db1), M, N.stream_ptr())
    N.check(rc, "dgppo_mlp_gi_bwd")


def gru1_head_fwd(gi, Wh, bhn, h0, W1, b1, W2, b2, hs, hprev, gates, u, out):
    """one GRU step (T = 1) + head Dense(s): out = (h' W1 + b1) [W2 + b2]; see dgppo_gru1_head_fwd in the header."""
    M = gi.shape[0]
    n_out = out.shape[1]
    N.expect_shape(gi, (M, 192), "gi"); N.expect_shape(Wh, (64, 192), "Wh"); N.expect_shape(bhn, (64,), "bhn")
    N.expect_shape(hs, (M, 64), "hs"); N.expect_shape(out, (M, n_out), "out")
    if h0 is not None:
        N.expect_shape(h0, (M, 64), "h0")
    if W2 is not None:
        N.expect_shape(W1, (64, 64), "W1"); N.expect_shape(b1, (64,), "b1")
        N.expect_shape(W2, (64, n_out), "W2"); N.expect_shape(b2, (n_out,), "b2")
    else:
        N.expect_shape(W1, (64, n_out), "W1"); N.expect_shape(b1, (n_out,), "b1")
    for t, shp, nm in ((hprev, (M, 64), "hprev"), (gates, (M, 256), "gates"), (u, (M, 64), "u")):
        if t is not None:
            N.expect_shape(t, shp, nm)
    FLOPS[0] += 2.0 * M * (64 * 192 + (64 * 64 + 64 * n_out if W2 is not None else 64 * n_out))
    rc = N.lib().dgppo_gru1_head_fwd(_p(gi), _p(Wh), _p(bhn), _p(h0), _p(W1), _p(b1), _p(W2), _p(b2), _p(hs), _p(hprev),
                                     _p(gates), _p(u), _p(out), M, n_out, N.stream_ptr())
    N.check(rc, "dgppo_gru1_head_fwd")


_WS: dict = {}   # (device index, stream handle) -> scratch tensor for dense_bwd_w's partial sums (caller-owned, see the header)


def _bwd_w_workspace(device, K: int, Ncol: int) -> torch.Tensor:
    lib = N.lib()
    lib.dgppo_dense_bwd_w_workspace_bytes.restype = C.c_int64
    need = int(lib.dgppo_dense_bwd_w_workspace_bytes(C.c_int32(K), C.c_int32(Ncol)))
    need = min(need, 64 << 20)                      # a smaller buffer only shrinks the grid
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _WS.get(key)
    if ws is None or ws.numel() * 4 < need:
        ws = torch.empty(need // 4, dtype=torch.float32, device=device)
        _WS[key] = ws
    return ws


class ReduceDesc(C.Structure):
    """struct dgppo_reduce_desc"""
    _fields_ = [("part", C.c_void_p), ("dW", C.c_void_p), ("db", C.c_void_p), ("part_stride", C.c_int32), ("slabs", C.c_int32),
                ("ldw", C.c_int32), ("K", C.c_int32), ("N", C.c_int32), ("pending", C.c_int32)]


class BwdWBatch:
    """Weight gradients of one backward pass with their second stage (slab reduction) deferred: every dense_bwd_w inside
    `with BwdWBatch(device, alloc):` launches only its partial-slab kernel into its own region of one workspace, and leaving
    the block flushes all pending reductions with one launch per 16 (dgppo_dense_bwd_w_reduce_batch).  `alloc(n_floats)`
    returns the caller's workspace of at least that size (an arena buffer: a move is visible to captured-graph checks)."""
    _active = {}        # (device index, stream handle) -> BwdWBatch

    def __init__(self, device, alloc):
        self.device, self.alloc = device, alloc
        self.key = None
        self.descs = []
        self.offset = 0                  # bytes used in the workspace
        self.ws = None

    def __enter__(self):
        self.key = (self.device.index, torch.cuda.current_stream(self.device).cuda_stream)
        if self.key in BwdWBatch._active:
            raise RuntimeError("BwdWBatch: nested batches on one stream")
        BwdWBatch._active[self.key] = self
        return self

    def region(self, nbytes: int):
        """a 256-byte aligned region of the workspace; regions of one batch never overlap"""
        nbytes = (nbytes + 255) & ~255
        if self.ws is None or self.ws.numel() * 4 < self.offset + nbytes:
            if self.descs:               # growing may move regions that pending reductions still read: flush them first
                self.flush()
            self.ws = self.alloc(max(self.offset + nbytes, 2 * (self.ws.numel() * 4 if self.ws is not None else 0)) // 4)
        view = self.ws[self.offset // 4:(self.offset + nbytes) // 4]
        self.offset += nbytes
        return view

    def flush(self):
        if self.descs:
            arr = (ReduceDesc * len(self.descs))(*self.descs)
            N.check(N.lib().dgppo_dense_bwd_w_reduce_batch(arr, C.c_int32(len(self.descs)), N.stream_ptr()),
                    "dgppo_dense_bwd_w_reduce_batch")
        self.descs = []
        self.offset = 0

    def __exit__(self, *exc):
        try:
            if exc[0] is None:
                self.flush()
        finally:
            BwdWBatch._active.pop(self.key, None)
        return False


def dense_bwd_w(X, dY, dW, db=None):
    """dW += X.T @ dY ; db += dY.sum(0)   (inside `with BwdWBatch(...)`: the slab reduction is deferred to the batch's flush)"""
    xp, ldx, M, K = _mat(X, "X")
    yp, ldy, My, Ncol = _mat(dY, "dY")
    wp, ldw, Kw, Nw = _mat(dW, "dW")
    if My != M or Kw != K or Nw != Ncol:
        raise ValueError(f"dense_bwd_w: shape mismatch X{tuple(X.shape)} dY{tuple(dY.shape)} dW{tuple(dW.shape)}")
    if db is not None:
        N.expect_shape(db, (Ncol,), "db")
    FLOPS[0] += 2.0 * M * K * Ncol
    batch = BwdWBatch._active.get((X.device.index, torch.cuda.current_stream(X.device).cuda_stream)) if X.is_cuda else None
    if batch is not None:
        lib = N.lib()
        lib.dgppo_dense_bwd_w_workspace_bytes.restype = C.c_int64
        need = min(int(lib.dgppo_dense_bwd_w_workspace_bytes(C.c_int32(K), C.c_int32(Ncol))), 64 << 20)
        ws = batch.region(need)
        d = ReduceDesc()
        rc = lib.dgppo_dense_bwd_w_deferred(xp, ldx, yp, ldy, wp, ldw, _p(db, "db"), M, K, Ncol, _p(ws, "workspace"),
                                            C.c_int64(ws.numel() * 4), C.byref(d), N.stream_ptr())
        N.check(rc, "dgppo_dense_bwd_w_deferred")
        if d.pending:
            batch.descs.append(d)
        return
    ws = _bwd_w_workspace(X.device, K, Ncol)
    rc = N.lib().dgppo_dense_bwd_w(xp, ldx, yp, ldy, wp, ldw, _p(db, "db"), M, K, Ncol, _p(ws, "workspace"),
                                   C.c_int64(ws.numel() * 4), N.stream_ptr())
    N.check(rc, "dgppo_dense_bwd_w")


def graph_feats(cfg: N.EnvCfg, agent, agent_se, agent_st, goal, obst, hits, hits_se, hits_st, env_ids, n_env, n_time,
                Xa, Xo, efeat, emask, Fp):
    """agent/hits are base tensors (any shape); strides are in floats."""
    G = n_env * n_time
    n, S = cfg.n_agents, cfg.fan_in
    n_other = cfg.num_nodes - 1 - n
    N.expect_shape(Xa, (G * n, Fp), "Xa")
    if n_other > 0:
        N.expect_shape(Xo, (G * n_other, Fp), "Xo")
    N.expect_shape(efeat, (G * n, S, 4), "efeat")
    N.expect_shape(emask, (G * n, S), "emask")
    # the records are addressed as data_ptr + env * se + time * st (a view's own outer strides are ignored): the innermost
    # record must be dense fp32 CUDA storage, and the storage must reach the last (env, time) the call touches
    for name, t_, se, st, inner in (("agent", agent, agent_se, agent_st, (n, cfg.state_dim)),
                                    ("hits", hits, hits_se, hits_st, (n, cfg.top_k, 2))):
        if t_ is None:
            continue
        if not (t_.is_cuda and t_.dtype == torch.float32):
            raise ValueError(f"graph_feats: {name} must be a float32 CUDA tensor")
        want, acc = [], 1
        for d in reversed(inner):
            want.append(acc)
            acc *= d
        if tuple(t_.shape[-len(inner):]) != tuple(inner) or list(t_.stride()[-len(inner):]) != want[::-1]:
            raise ValueError(f"graph_feats: the trailing {inner} block of {name} must be dense (got shape {tuple(t_.shape)}, "
                             f"strides {t_.stride()})")
        avail = t_.untyped_storage().nbytes() // 4 - t_.storage_offset()
        if env_ids is None and (n_env - 1) * se + (n_time - 1) * st + acc > avail:
            raise ValueError(f"graph_feats: {name}: the strides reach beyond its storage")
    rc = N.lib().dgppo_graph_feats(
        C.byref(cfg), C.c_void_p(agent.data_ptr()), C.c_int64(agent_se), C.c_int64(agent_st), _p(goal, "goal"),
        _p(obst, "obst") if (obst is not None and not cfg.is_lidar) else C.c_void_p(0),
        C.c_void_p(hits.data_ptr()) if hits is not None else C.c_void_p(0), C.c_int64(hits_se), C.c_int64(hits_st),
        _p(env_ids, "env_ids", torch.int32), C.c_int32(n_env), C.c_int32(n_time), _p(Xa, "Xa"), _p(Xo, "Xo"),
        _p(efeat, "efeat"), _p(emask, "emask"), C.c_int32(Fp), N.stream_ptr())
    N.check(rc, "dgppo_graph_feats")


def attn_fwd(cfg, F, H, Kp, qt, Xa, Xo, efeat, emask, zcat, attn, G):
    n, S = cfg.n_agents, cfg.fan_in
    N.expect_shape(qt, (G * n, H * F), "qt")
    N.expect_shape(Xa, (G * n, F), "Xa")
    N.expect_shape(zcat, (G * n, Kp), "zcat")
    if attn is not None:                                     # None: inference, the attention weights are not kept
        N.expect_shape(attn, (G * n, S, H), "attn")
    FLOPS[0] += 2.0 * G * n * H * S * (2 * F + 4)            # logits (F) + aggregation of [x_s | e] (F + 4) per (agent, head, slot)
    rc = N.lib().dgppo_attn_fwd(C.byref(cfg), F, H, Kp, _p(qt), _p(Xa), _p(Xo), _p(efeat), _p(emask), _p(zcat), _p(attn),
                                G, N.stream_ptr())
    N.check(rc, "dgppo_attn_fwd")


def attn_bwd(cfg, F, H, Kp, dzcat, attn, qt, Xa, Xo, efeat, dqt, dXa, dXo, G, relu_xo: bool = False):
    n = cfg.n_agents
    N.expect_shape(dzcat, (G * n, Kp), "dzcat")
    N.expect_shape(dqt, (G * n, H * F), "dqt")
    if dXa is not None:
        N.expect_shape(dXa, (G * n, F), "dXa")
    FLOPS[0] += 4.0 * G * n * H * cfg.fan_in * (2 * F + 4)   # dA, dL -> dqt, dXs: twice the forward contractions
    rc = N.lib().dgppo_attn_bwd(C.byref(cfg), F, H, Kp, _p(dzcat), _p(attn), _p(qt), _p(Xa), _p(Xo), _p(efeat), _p(dqt),
                                _p(dXa), _p(dXo), int(relu_xo), G, N.stream_ptr())
    N.check(rc, "dgppo_attn_bwd")


def attn_xo_supported(cfg, F, H, Kp) -> bool:
    """does the topology have an attention kernel that recomputes the other nodes' rows (dgppo_attn_fwd_xo / _bwd_xo)?"""
    return bool(N.lib().dgppo_attn_xo_supported(C.byref(cfg), F, H, Kp))


def _wo_ptr(Wo):
    if not Wo.is_cuda or Wo.dtype != torch.float32:
        raise TypeError("Wo must be a float32 GPU tensor")
    return C.c_void_p(Wo.data_ptr())


def _wo_ld(Wo):
    if Wo.dim() != 2 or Wo.shape[0] != 8 or Wo.shape[1] < 32 or Wo.stride(1) != 1:
        raise ValueError(f"Wo must be [8, >= 32] with contiguous rows, got {tuple(Wo.shape)} strides {Wo.stride()}")
    return int(Wo.stride(0))


def attn_fwd_xo(cfg, F, H, Kp, qt, Xa, Xo_raw, Wo, bo, efeat, emask, zcat, attn, G):
    """attn_fwd with Xo = relu(Xo_raw Wo + bo) recomputed in the kernel (Xo_raw [G*n_other, 8], Wo [8, 32] rows of a wider matrix)."""
    n, S = cfg.n_agents, cfg.fan_in
    N.expect_shape(qt, (G * n, H * F), "qt")
    N.expect_shape(Xa, (G * n, F), "Xa")
    N.expect_shape(zcat, (G * n, Kp), "zcat")
    N.expect_shape(Xo_raw, (G * (cfg.num_nodes - 1 - n), 8), "Xo_raw")
    N.expect_shape(bo, (32,), "bo")
    if attn is not None:
        N.expect_shape(attn, (G * n, S, H), "attn")
    FLOPS[0] += 2.0 * G * n * H * S * (2 * F + 4) + 2.0 * G * (cfg.num_nodes - 1 - n) * 8 * 32
    rc = N.lib().dgppo_attn_fwd_xo(C.byref(cfg), F, H, Kp, _p(qt), _p(Xa), _p(Xo_raw), _wo_ptr(Wo),
                                   _wo_ld(Wo), _p(bo), _p(efeat), _p(emask), _p(zcat), _p(attn), G, N.stream_ptr())
    N.check(rc, "dgppo_attn_fwd_xo")


def attn_bwd_xo(cfg, F, H, Kp, dzcat, attn, qt, Xa, Xo_raw, Wo, bo, efeat, dqt, dXa, dXo, G, relu_xo: bool = False):
    n = cfg.n_agents
    N.expect_shape(dzcat, (G * n, Kp), "dzcat")
    N.expect_shape(dqt, (G * n, H * F), "dqt")
    N.expect_shape(Xo_raw, (G * (cfg.num_nodes - 1 - n), 8), "Xo_raw")
    if dXa is not None:
        N.expect_shape(dXa, (G * n, F), "dXa")
    FLOPS[0] += 4.0 * G * n * H * cfg.fan_in * (2 * F + 4) + 2.0 * G * (cfg.num_nodes - 1 - n) * 8 * 32
    rc = N.lib().dgppo_attn_bwd_xo(C.byref(cfg), F, H, Kp, _p(dzcat), _p(attn), _p(qt), _p(Xa), _p(Xo_raw),
                                   _wo_ptr(Wo), _wo_ld(Wo), _p(bo), _p(efeat), _p(dqt), _p(dXa), _p(dXo),
                                   int(relu_xo), G, N.stream_ptr())
    N.check(rc, "dgppo_attn_bwd_xo")


def attn_xo_workspace_floats(G: int) -> int:
    return int(N.lib().dgppo_attn_xo_workspace_bytes(int(G))) // 4


def attn_bwd_xo_dw(cfg, F, H, Kp, dzcat, attn, qt, Xa, Xo_raw, Wo, bo, efeat, dqt, dXa, dWo, dbo, workspace, G):
    """attn_bwd_xo that consumes the gradient of the recomputed rows: dWo [8, >= 32 wide rows] += Xo_raw^T (relu' * dXo),
    dbo [32] += its column sums; workspace: attn_xo_workspace_floats(G) floats."""
    n = cfg.n_agents
    N.expect_shape(dzcat, (G * n, Kp), "dzcat")
    N.expect_shape(dqt, (G * n, H * F), "dqt")
    N.expect_shape(dXa, (G * n, F), "dXa")
    N.expect_shape(Xo_raw, (G * (cfg.num_nodes - 1 - n), 8), "Xo_raw")
    N.expect_shape(dbo, (32,), "dbo")
    FLOPS[0] += 4.0 * G * n * H * cfg.fan_in * (2 * F + 4) + 4.0 * G * (cfg.num_nodes - 1 - n) * 8 * 32
    rc = N.lib().dgppo_attn_bwd_xo_dw(C.byref(cfg), F, H, Kp, _p(dzcat), _p(attn), _p(qt), _p(Xa), _p(Xo_raw), _wo_ptr(Wo), _wo_ld(Wo),
                                      _p(bo), _p(efeat), _p(dqt), _p(dXa), _wo_ptr(dWo), _wo_ld(dWo), _p(dbo), _p(workspace),
                                      C.c_int64(workspace.numel() * 4), G, N.stream_ptr())
    N.check(rc, "dgppo_attn_bwd_xo_dw")


def gnn_prep(Wq, bq, Wk, Wv, bv, We, Wu, Mcat, cvec, Wout, F, Fp, D, H, Kp):
    rc = N.lib().dgppo_gnn_prep(_p(Wq), _p(bq), _p(Wk), _p(Wv), _p(bv), _p(We), _p(Wu), _p(Mcat), _p(cvec), _p(Wout),
                                F, Fp, D, H, Kp, N.stream_ptr())
    N.check(rc, "dgppo_gnn_prep")


def gnn_unprep(dMcat, dcvec, dWout, Wq, bq, Wk, dWq, dbq, dWk, dWv, dbv, dWe, dWu, F, Fp, D, H, Kp):
    rc = N.lib().dgppo_gnn_unprep(_p(dMcat), _p(dcvec), _p(dWout), _p(Wq), _p(bq), _p(Wk), _p(dWq), _p(dbq), _p(dWk),
                                  _p(dWv), _p(dbv), _p(dWe), _p(dWu), F, Fp, D, H, Kp, N.stream_ptr())
    N.check(rc, "dgppo_gnn_unprep")


def ln_relu_fwd(x, gamma, beta, y, stats):
    M = x.shape[0]
    N.expect_shape(x, (M, 64), "x")
    N.expect_shape(y, (M, 64), "y")
    rc = N.lib().dgppo_ln_relu_fwd(_p(x), _p(gamma), _p(beta), _p(y), _p(stats), M, N.stream_ptr())
    N.check(rc, "dgppo_ln_relu_fwd")


def ln_relu_bwd(x, y, stats, gamma, dy, dx, dgamma, dbeta):
    M = x.shape[0]
    rc = N.lib().dgppo_ln_relu_bwd(_p(x), _p(y), _p(stats), _p(gamma), _p(dy), _p(dx), _p(dgamma), _p(dbeta), M,
                                   N.stream_ptr())
    N.check(rc, "dgppo_ln_relu_bwd")


def gru_fwd(gi, Wh, bhn, h0, hs, hprev, gates, n_seq, T, n_inner):
    rows = n_seq * T
    N.expect_shape(gi, (rows, 192), "gi")
    N.expect_shape(hs, (rows, 64), "hs")
    N.expect_shape(Wh, (64, 192), "Wh")
    if h0 is not None:
        N.expect_shape(h0, (n_seq, 64), "h0")
    FLOPS[0] += 2.0 * rows * 64 * 192
    rc = N.lib().dgppo_gru_fwd(_p(gi), _p(Wh), _p(bhn), _p(h0), _p(hs), _p(hprev), _p(gates), n_seq, T, n_inner,
                               N.stream_ptr())
    N.check(rc, "dgppo_gru_fwd")


def gru_bwd(dhs, Wh, hprev, gates, dgi, dgh, n_seq, T, n_inner):
    rows = n_seq * T
    N.expect_shape(dhs, (rows, 64), "dhs")
    N.expect_shape(dgi, (rows, 192), "dgi")
    N.expect_shape(dgh, (rows, 192), "dgh")
    FLOPS[0] += 2.0 * rows * 64 * 192
    rc = N.lib().dgppo_gru_bwd(_p(dhs), _p(Wh), _p(hprev), _p(gates), _p(dgi), _p(dgh), n_seq, T, n_inner, N.stream_ptr())
    N.check(rc, "dgppo_gru_bwd")


def policy_head(ms, eps, action_in, action, log_pi, entropy, n_agents, mode, log_pi_old=None, adv=None, dms=None,
                stats=None, clip_eps=0.25, coef_ent=0.01):
    rows = ms.shape[0]
    N.expect_shape(ms, (rows, 4), "ms")
    rc = N.lib().dgppo_policy_head(_p(ms), _p(eps), _p(action_in), _p(action), _p(log_pi), _p(entropy), rows, n_agents,
                                   mode, _p(log_pi_old), _p(adv), _p(dms), _p(stats), C.c_float(clip_eps),
                                   C.c_float(coef_ent), N.stream_ptr())
    N.check(rc, "dgppo_policy_head")


def value_loss(v, target, dv, stats):
    rc = N.lib().dgppo_value_loss(_p(v), _p(target), _p(dv), _p(stats), v.numel(), N.stream_ptr())
    N.check(rc, "dgppo_value_loss")


def mean_agents(x, y, G, n, D, backward=False, relu_mask=None):
    """backward: False/0 mean, True/1 gradient of the mean, 2 broadcast-add (see the header)"""
    rc = N.lib().dgppo_mean_agents(_p(x), _p(y), G, n, D, int(backward), _p(relu_mask), N.stream_ptr())
    N.check(rc, "dgppo_mean_agents")


def relu_bwd(dy, y):
    rc = N.lib().dgppo_relu_bwd(_p(dy), _p(y), C.c_int64(dy.numel()), N.stream_ptr())
    N.check(rc, "dgppo_relu_bwd")


def lstm_fwd(zi, Wh, bh, c0, h0, cs, hs, cprev, hprev, gates, n_seq, T, n_inner):
    rows = n_seq * T
    N.expect_shape(zi, (rows, 256), "zi"); N.expect_shape(Wh, (64, 256), "Wh"); N.expect_shape(bh, (256,), "bh")
    N.expect_shape(cs, (rows, 64), "cs"); N.expect_shape(hs, (rows, 64), "hs")
    for t, nm in ((c0, "c0"), (h0, "h0")):
        if t is not None:
            N.expect_shape(t, (n_seq, 64), nm)
    FLOPS[0] += 2.0 * rows * 64 * 256
    rc = N.lib().dgppo_lstm_fwd(_p(zi), _p(Wh), _p(bh), _p(c0), _p(h0), _p(cs), _p(hs), _p(cprev), _p(hprev), _p(gates),
                                n_seq, T, n_inner, N.stream_ptr())
    N.check(rc, "dgppo_lstm_fwd")


def lstm_bwd(dhs, Wh, cprev, gates, dz, n_seq, T, n_inner):
    rows = n_seq * T
    N.expect_shape(dhs, (rows, 64), "dhs"); N.expect_shape(dz, (rows, 256), "dz")
    FLOPS[0] += 2.0 * rows * 64 * 256
    rc = N.lib().dgppo_lstm_bwd(_p(dhs), _p(Wh), _p(cprev), _p(gates), _p(dz), n_seq, T, n_inner, N.stream_ptr())
    N.check(rc, "dgppo_lstm_bwd")
