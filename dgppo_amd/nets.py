"""Host-side composition of the HIP building blocks into the reference's three networks
(PolicyNet+TanhNormal: dgppo/algo/module/policy.py:20-78; RStateFn / DecRStateFn: dgppo/algo/module/value.py:15-79).

Parameters live in ONE flat fp32 device buffer per network (so grad-norm / clip / Adam / all-reduce are single
kernels); `layout` maps flax-style names to (offset, shape).  Torch tensors are storage only."""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import _native as N
from . import ops_nn as K

H_HEADS = 3
MSG_DIM = 32
OUT_DIM = 64
HID = 64


def _ceil4(x):
    return (x + 3) // 4 * 4


class Layout:
    def __init__(self):
        self.entries: Dict[str, Tuple[int, Tuple[int, ...]]] = {}
        self.size = 0

    def add(self, name, *shape):
        n = int(np.prod(shape))
        # keep every segment 16-byte aligned (float4 loads, MFMA-friendly)
        self.size = (self.size + 3) // 4 * 4
        self.entries[name] = (self.size, tuple(shape))
        self.size += n

    def view(self, flat: torch.Tensor, name: str) -> torch.Tensor:
        off, shape = self.entries[name]
        return flat[off:off + math.prod(shape)].view(shape)


def make_layout(kind: str, node_dim: int, gnn_layers: int, n_out: int, rnn: str = "gru", rnn_layers: int = 1) -> Layout:
    """kind: 'policy' | 'Vl' | 'Vh' | 'Vhg' (= DecRStateFn(use_global_info=True): the head sees [x_i | mean_j x_j],
    value.py:66-68, so its first Dense is 128 wide).  Dense kernels are [in, out] like flax."""
    L = Layout()
    f = node_dim
    for l in range(gnn_layers):
        d = OUT_DIM if l == gnn_layers - 1 else MSG_DIM
        hd = H_HEADS * d
        for nm, shp in (("Wq", (f, hd)), ("bq", (hd,)), ("Wk", (f, hd)), ("bk", (hd,)), ("Wv", (f, hd)), ("bv", (hd,)),
                        ("We", (4, hd)), ("Wu", (f, d)), ("bu", (d,))):
            L.add(f"gnn{l}.{nm}", *shp)
        f = d
    for i in (1, 2):
        L.add(f"mlp.W{i}", 2 * HID if (kind == "Vhg" and i == 1) else HID, HID)
        L.add(f"mlp.b{i}", HID)
        L.add(f"mlp.g{i}", HID)
        L.add(f"mlp.be{i}", HID)
    # recurrent stack (dgppo/nn/rnn.py:14-30): layer 0 keeps the historical names, layer l >= 1 is "gru{l}.*"; rnn == "none"
    # (--no-rnn: PolicyNet / ValueNet without a cell, policy.py:31-32, value.py:38-39) has no recurrent parameters
    assert rnn in ("gru", "lstm", "none"), rnn
    for l in range(rnn_layers if rnn == "lstm" else 0):   # flax LSTMCell: input Denses without bias, hidden Denses with
        L.add(f"lstm{l}.Wi", HID, 4 * HID)   # ii | if | ig | io
        L.add(f"lstm{l}.Wh", HID, 4 * HID)   # hi | hf | hg | ho
        L.add(f"lstm{l}.bh", 4 * HID)
    for l in range(rnn_layers if rnn == "gru" else 0):
        pre = "gru" if l == 0 else f"gru{l}"
        L.add(f"{pre}.Wi", HID, 3 * HID)   # ir | iz | in
        L.add(f"{pre}.bi", 3 * HID)
        L.add(f"{pre}.Wh", HID, 3 * HID)   # hr | hz | hn
        L.add(f"{pre}.bhn", HID)
    if kind == "policy":
        L.add("head.Ws", HID, HID)      # ScaleHid
        L.add("head.bs", HID)
        L.add("head.Wms", HID, 2 * n_out)  # OutputDenseMean | OutputDenseStdTrans
        L.add("head.bms", 2 * n_out)
    else:
        L.add("head.Wo", HID, n_out)
        L.add("head.bo", n_out)
    return L


class Arena:
    """named scratch tensors, reused across calls (sized for the largest request seen)."""

    def __init__(self, device):
        self.device = device
        self.bufs: Dict[str, torch.Tensor] = {}
        self.generation = 0          # bumped whenever a buffer is re-allocated (moves): captured HIP graphs check it before replay

    def get(self, name: str, *shape, dtype=torch.float32, zero=False) -> torch.Tensor:
        n = math.prod(shape)
        b = self.bufs.get(name)
        if b is None or b.numel() < n or b.dtype != dtype:
            if b is not None:
                self.generation += 1     # an existing buffer moves: pointers baked into captured graphs are stale
            b = torch.empty(max(n, 1), dtype=dtype, device=self.device)
            self.bufs[name] = b
        v = b[:n].view(shape)
        if zero:
            v.zero_()
        return v


class GraphFeats:
    """dense per-graph features shared by the three networks (output of dgppo_graph_feats)."""

    def __init__(self, cfg: N.EnvCfg, G: int, arena: Arena, tag: str):
        self.cfg, self.G = cfg, G
        n, S = cfg.n_agents, cfg.fan_in
        self.n_other = cfg.num_nodes - 1 - n
        self.Fp = 8
        self.Xa = arena.get(f"{tag}.Xa0", G * n, self.Fp)
        self.Xo = arena.get(f"{tag}.Xo0", max(G * self.n_other, 1), self.Fp)[:G * self.n_other]
        self.efeat = arena.get(f"{tag}.efeat", G * n, S, 4)
        self.emask = arena.get(f"{tag}.emask", G * n, S)

    def compute(self, agent, agent_se, agent_st, goal, obst, hits, hits_se, hits_st, env_ids, n_env, n_time):
        assert n_env * n_time == self.G
        K.graph_feats(self.cfg, agent, agent_se, agent_st, goal, obst, hits, hits_se, hits_st, env_ids, n_env, n_time,
                      self.Xa, self.Xo if self.n_other > 0 else None, self.efeat, self.emask, self.Fp)
        return self


class Net:
    """One network = flat params + flat grads + prepared GNN weights + forward/backward over a batch of graphs."""

    def __init__(self, kind: str, cfg: N.EnvCfg, gnn_layers: int, n_out: int, device, grads: Optional[torch.Tensor] = None,
                 rnn: str = "gru", rnn_layers: int = 1):
        """grads: optional caller-owned flat gradient buffer (a slice of the engine's [g_policy | g_Vl | g_Vh | scalars]
        buffer, so that the data-parallel update all-reduces ALL gradients with one collective).
        rnn / rnn_layers: "gru" x L stacked cells (dgppo/nn/rnn.py:14-30; the carry of a row is [L * 64]) or "none"."""
        assert kind in ("policy", "Vl", "Vh", "Vhg")
        self.kind, self.cfg, self.gnn_layers, self.n_out, self.device = kind, cfg, gnn_layers, n_out, device
        self.rnn, self.rnn_layers = rnn, (rnn_layers if rnn != "none" else 0)
        # packed carry of one row: GRU [h_0 | h_1 | ...]; LSTM [c_0 | h_0 | c_1 | h_1 | ...] (the reference stacks (c, h) as
        # carries 0 and 1 of a layer, rnn.py:23-24); without a cell 64 zeros pass through
        self.carry_dim = HID * max(self.rnn_layers, 1) * (2 if rnn == "lstm" else 1)
        self.layout = make_layout(kind, cfg.node_dim, gnn_layers, n_out, rnn, rnn_layers)
        self.params = torch.zeros(self.layout.size, device=device)
        if grads is None:
            grads = torch.zeros(self.layout.size, device=device)
        assert grads.shape == (self.layout.size,) and grads.is_contiguous() and grads.data_ptr() % 16 == 0
        self.grads = grads
        self.arena = Arena(device)
        self.ws_arena = Arena(device)      # backward-only workspace (grows during the first update; the rollout graphs, which
                                           # hold pointers into `arena`, must not be invalidated by that)
        self._views: Dict[tuple, torch.Tensor] = {}
        # per-layer dims: F (true input width), Fp (padded), D, Kp
        self.dims = []
        f = cfg.node_dim
        for l in range(gnn_layers):
            d = OUT_DIM if l == gnn_layers - 1 else MSG_DIM
            fp = 8 if l == 0 else f
            kp = _ceil4(fp + H_HEADS * (fp + 4) + 1)
            self.dims.append((f, fp, d, kp))
            f = d
        # prepared weights (+ their grads) in one flat buffer each
        self.prep_layout = Layout()
        for l, (f, fp, d, kp) in enumerate(self.dims):
            self.prep_layout.add(f"gnn{l}.Mcat", fp, H_HEADS * fp)
            self.prep_layout.add(f"gnn{l}.cvec", H_HEADS * fp)
            self.prep_layout.add(f"gnn{l}.Wout", kp, d)
        self.prep = torch.zeros(self.prep_layout.size, device=device)
        self.prep_grads = torch.zeros(self.prep_layout.size, device=device)

    # ---- parameter access ------------------------------------------------------------------------------------------
    # views into the flat buffers; the buffers never move (in-place updates only), so the views are built once
    def _cached(self, kind: str, layout, flat, name):
        key = (kind, name)
        v = self._views.get(key)
        if v is None:
            v = self._views[key] = layout.view(flat, name)
        return v

    def p(self, name):
        return self._cached("p", self.layout, self.params, name)

    def g(self, name):
        return self._cached("g", self.layout, self.grads, name)

    def pp(self, name):
        return self._cached("pp", self.prep_layout, self.prep, name)

    def pg(self, name):
        return self._cached("pg", self.prep_layout, self.prep_grads, name)

    def prepare(self):
        """params -> prepared GNN weights (call after every parameter change)."""
        for l, (f, fp, d, kp) in enumerate(self.dims):
            q = lambda nm: self.p(f"gnn{l}.{nm}")
            K.gnn_prep(q("Wq"), q("bq"), q("Wk"), q("Wv"), q("bv"), q("We"), q("Wu"), self.pp(f"gnn{l}.Mcat"),
                       self.pp(f"gnn{l}.cvec"), self.pp(f"gnn{l}.Wout"), f, fp, d, H_HEADS, kp)

    def _xo_fusable(self, l: int) -> bool:
        """may layer l + 1's attention recompute the other nodes' rows relu(Xo_l Wout_l[:fp] + bu_l) itself?  Only from the
        8-wide raw features (l = 0) into a LAST layer (a third layer would need the rows as a tensor), and only where the C ABI
        has such a kernel for the topology (dgppo_attn_xo_supported)."""
        if l != 0 or self.gnn_layers != 2:
            return False
        key = "_xo_ok"
        if not hasattr(self, key):
            f1, fp1, d1, kp1 = self.dims[1]
            setattr(self, key, self.dims[0][1] == 8 and self.dims[0][2] == 32 and fp1 == 32 and
                    K.attn_xo_supported(self.cfg, fp1, H_HEADS, kp1))
        return getattr(self, key)

    def zero_grads(self):
        # (the prepared-weight gradients are zero between backward passes: backward() clears them after mapping them back)
        self.grads.zero_()

    # ---- forward ---------------------------------------------------------------------------------------------------
    def forward(self, feats: GraphFeats, n_seq: int, T: int, h0: Optional[torch.Tensor], tag: str = "f",
                hs_out: Optional[torch.Tensor] = None, train: bool = True):
        """Trunk + GRU + output Dense(s).  Graphs are ordered (group, time): G = (n_seq / n_inner) * T.
        Returns a dict of activations (views into this net's arena, valid until the next forward with the same tag)."""
        cfg, G, A = self.cfg, feats.G, self.arena
        n = cfg.n_agents
        R = G * n
        Ro = G * feats.n_other
        act = {"feats": feats, "G": G, "n_seq": n_seq, "T": T}
        Xa, Xo = feats.Xa, feats.Xo
        xo_fused = None
        for l, (f, fp, d, kp) in enumerate(self.dims):
            qt = A.get(f"{tag}.qt{l}", R, H_HEADS * fp)
            K.dense_fwd(Xa, self.pp(f"gnn{l}.Mcat"), self.pp(f"gnn{l}.cvec"), qt)
            zcat = A.get(f"{tag}.zcat{l}", R, kp)
            attn = A.get(f"{tag}.attn{l}", R, cfg.fan_in, H_HEADS) if train else None    # only the backward reads the weights
            if xo_fused is not None:      # other nodes' rows recomputed in the kernel from the raw features (see below)
                K.attn_fwd_xo(cfg, fp, H_HEADS, kp, qt, Xa, xo_fused[0], xo_fused[1], xo_fused[2], feats.efeat, feats.emask, zcat, attn, G)
            else:
                K.attn_fwd(cfg, fp, H_HEADS, kp, qt, Xa, Xo if Ro > 0 else None, feats.efeat, feats.emask, zcat, attn, G)
            act[f"xo_fused{l}"] = xo_fused
            xo_fused = None
            Xa_n = A.get(f"{tag}.Xa{l + 1}", R, d)
            K.dense_fwd(zcat, self.pp(f"gnn{l}.Wout"), self.p(f"gnn{l}.bu"), Xa_n, act=1)
            act[f"Xa{l}"], act[f"Xo{l}"], act[f"qt{l}"], act[f"zcat{l}"], act[f"attn{l}"] = Xa, Xo, qt, zcat, attn
            if l < self.gnn_layers - 1 and Ro > 0:
                # goals / hits / obstacles receive no messages: relu(W_u x + b_u)  (gnn.py:109-111 with aggr = 0)
                if self._xo_fusable(l):
                    # ... and when the next layer is the last one, its attention kernels recompute these rows from the 8 raw
                    # features instead of reading 32-wide rows that are written here and read twice more (forward, backward)
                    xo_fused = (Xo, self.pp(f"gnn{l}.Wout")[:fp], self.p(f"gnn{l}.bu"))
                    Xo = None
                else:
                    Xo_n = A.get(f"{tag}.Xo{l + 1}", Ro, d)
                    K.dense_fwd(Xo, self.pp(f"gnn{l}.Wout")[:fp], self.p(f"gnn{l}.bu"), Xo_n, act=1)
                    Xo = Xo_n
            Xa = Xa_n
        act[f"Xa{self.gnn_layers}"] = Xa
        if self.kind == "Vl":  # RStateFn: mean over agents (value.py:33)
            pooled = A.get(f"{tag}.pool", G, OUT_DIM)
            K.mean_agents(Xa, pooled, G, n, OUT_DIM)
            x, Rh, n_inner = pooled, G, 1
        else:
            x, Rh, n_inner = Xa, R, n
        if self.kind == "Vhg":   # [x_i | mean over the graph's agents, tiled] (value.py:66-68): data movement + one mean kernel
            pooled = A.get(f"{tag}.pool", G, OUT_DIM)
            K.mean_agents(Xa, pooled, G, n, OUT_DIM)
            xcat = A.get(f"{tag}.xcat", R, 2 * OUT_DIM)
            xcat[:, :OUT_DIM].copy_(Xa)
            xcat.view(G, n, 2 * OUT_DIM)[:, :, OUT_DIM:].copy_(pooled.view(G, 1, OUT_DIM).expand(G, n, OUT_DIM))
            x = xcat
        act["Rh"], act["n_inner"], act["mlp_in"] = Rh, n_inner, x
        gi = A.get(f"{tag}.gi", Rh, 3 * HID)
        if self.kind == "Vhg" or self.rnn != "gru":
            # 128-wide first Dense: the separate Dense / LayerNorm+ReLU kernels (the fused trunk kernel is 64-wide)
            sv = {nm: A.get(f"{tag}.{nm}", Rh, w) for nm, w in (("p1", HID), ("y1", HID), ("st1", 2), ("p2", HID), ("y2", HID), ("st2", 2))}
            K.dense_fwd(x, self.p("mlp.W1"), self.p("mlp.b1"), sv["p1"])
            K.ln_relu_fwd(sv["p1"], self.p("mlp.g1"), self.p("mlp.be1"), sv["y1"], sv["st1"])
            K.dense_fwd(sv["y1"], self.p("mlp.W2"), self.p("mlp.b2"), sv["p2"])
            K.ln_relu_fwd(sv["p2"], self.p("mlp.g2"), self.p("mlp.be2"), sv["y2"], sv["st2"])
            if self.rnn == "gru":
                K.dense_fwd(sv["y2"], self.p("gru.Wi"), self.p("gru.bi"), gi)
            act.update(sv)
        else:
            # MLP trunk (2 x Dense -> LayerNorm -> ReLU) + GRU input projection: one fused kernel (nn_fused.hip)
            saves = None
            if train:
                saves = tuple(A.get(f"{tag}.{nm}{i}", Rh, w) for i in (1, 2) for nm, w in (("p", HID), ("y", HID), ("st", 2)))
                for i in (1, 2):
                    act[f"p{i}"], act[f"y{i}"], act[f"st{i}"] = saves[3 * (i - 1):3 * i]
            K.mlp_gi_fwd(x, self.p("mlp.W1"), self.p("mlp.b1"), self.p("mlp.g1"), self.p("mlp.be1"), self.p("mlp.W2"),
                         self.p("mlp.b2"), self.p("mlp.g2"), self.p("mlp.be2"), self.p("gru.Wi"), self.p("gru.bi"), gi, saves)
        assert n_seq * T == Rh, (n_seq, T, Rh)
        L, CD = self.rnn_layers, self.carry_dim
        if h0 is not None:
            assert tuple(h0.shape) == (n_seq, CD), (tuple(h0.shape), n_seq, CD)
        if hs_out is not None:
            assert tuple(hs_out.shape) == (Rh, CD), (tuple(hs_out.shape), Rh, CD)
        act["gi"] = gi
        simple = (self.rnn == "gru" and L == 1)           # the reference default: every fused kernel applies
        if simple:
            hs = hs_out if hs_out is not None else A.get(f"{tag}.hs", Rh, HID)
            hprev = A.get(f"{tag}.hprev", Rh, HID) if train else None
            gates = A.get(f"{tag}.gates", Rh, 4 * HID) if train else None
            feat = hs
            act["hs"], act["hprev"], act["gates"] = hs, hprev, gates
        # one GRU step: the step and the head Dense(s) are row-local -> fused kernel.  Measured (MI355X): policy rollout
        # shape 23.5 vs 34.2 us; for the one-layer value heads at pre-pass sizes the plain GRU kernel + Dense is faster
        # (261 vs 350 us at 524 288 rows), so those keep the separate kernels.
        fused_tail = (simple and T == 1 and self.kind == "policy")
        if simple and not fused_tail:
            K.gru_fwd(gi, self.p("gru.Wh"), self.p("gru.bhn"), h0, hs, hprev, gates, n_seq, T, n_inner)
        if self.rnn == "none":
            # no cell: the MLP output is the feature and the carry passes through unchanged (policy.py:29-33)
            feat = act["y2"]
            if hs_out is not None:
                if h0 is not None and T == 1:
                    hs_out.copy_(h0)
                else:
                    hs_out.zero_()
        elif self.rnn == "lstm":
            # LSTM cells (rnn.py:22-24): per layer an input projection (Dense without bias) and the scan kernel; the carry of
            # a row packs (c_l, h_l) per layer
            x_l = act["y2"]
            act["stack"] = []
            for l in range(L):
                zi = A.get(f"{tag}.zi{l}", Rh, 4 * HID)
                for half in (slice(0, 2 * HID), slice(2 * HID, 4 * HID)):        # the Dense kernels take N <= 192: two 128-wide halves
                    K.dense_fwd(x_l, self.p(f"lstm{l}.Wi")[:, half], None, zi[:, half])
                c0_l = h0_l = None
                if h0 is not None:
                    c0_l, h0_l = A.get(f"{tag}.c0_{l}", n_seq, HID), A.get(f"{tag}.h0_{l}", n_seq, HID)
                    c0_l.copy_(h0[:, (2 * l) * HID:(2 * l + 1) * HID])
                    h0_l.copy_(h0[:, (2 * l + 1) * HID:(2 * l + 2) * HID])
                cs_l, hs_l = A.get(f"{tag}.cs{l}", Rh, HID), A.get(f"{tag}.hs{l}", Rh, HID)
                cprev_l = A.get(f"{tag}.cprev{l}", Rh, HID) if train else None
                hprev_l = A.get(f"{tag}.hprev{l}", Rh, HID) if train else None
                gates_l = A.get(f"{tag}.gates{l}", Rh, 4 * HID) if train else None
                K.lstm_fwd(zi, self.p(f"lstm{l}.Wh"), self.p(f"lstm{l}.bh"), c0_l, h0_l, cs_l, hs_l, cprev_l, hprev_l, gates_l,
                           n_seq, T, n_inner)
                if hs_out is not None:
                    hs_out[:, (2 * l) * HID:(2 * l + 1) * HID].copy_(cs_l)
                    hs_out[:, (2 * l + 1) * HID:(2 * l + 2) * HID].copy_(hs_l)
                act["stack"].append(dict(x=x_l, hs=hs_l, cprev=cprev_l, hprev=hprev_l, gates=gates_l))
                x_l = hs_l
            feat = x_l
        elif not simple:
            # stacked cells (dgppo/nn/rnn.py:17-29): layer l consumes the output sequence of layer l-1; the carry of a row
            # is [h_0 | h_1 | ...].  Separate kernels per layer (input projection, scan); slices of the packed carry are
            # copied to contiguous buffers for the scan kernels.
            x_l = None
            act["stack"] = []
            for l in range(L):
                pre = "gru" if l == 0 else f"gru{l}"
                gi_l = gi if l == 0 else A.get(f"{tag}.gi{l}", Rh, 3 * HID)
                if l > 0:
                    K.dense_fwd(x_l, self.p(f"{pre}.Wi"), self.p(f"{pre}.bi"), gi_l)
                h0_l = None
                if h0 is not None:
                    h0_l = A.get(f"{tag}.h0_{l}", n_seq, HID)
                    h0_l.copy_(h0[:, l * HID:(l + 1) * HID])
                hs_l = A.get(f"{tag}.hs{l}", Rh, HID)
                hprev_l = A.get(f"{tag}.hprev{l}", Rh, HID) if train else None
                gates_l = A.get(f"{tag}.gates{l}", Rh, 4 * HID) if train else None
                K.gru_fwd(gi_l, self.p(f"{pre}.Wh"), self.p(f"{pre}.bhn"), h0_l, hs_l, hprev_l, gates_l, n_seq, T, n_inner)
                if hs_out is not None:
                    hs_out[:, l * HID:(l + 1) * HID].copy_(hs_l)
                act["stack"].append(dict(x=x_l, gi=gi_l, hs=hs_l, hprev=hprev_l, gates=gates_l))
                x_l = hs_l
            feat = x_l
        act["feat"] = feat
        if self.kind == "policy":
            u = A.get(f"{tag}.u", Rh, HID) if (train or not fused_tail) else None
            ms = A.get(f"{tag}.ms", Rh, 4)
            if fused_tail:
                K.gru1_head_fwd(gi, self.p("gru.Wh"), self.p("gru.bhn"), h0, self.p("head.Ws"), self.p("head.bs"),
                                self.p("head.Wms"), self.p("head.bms"), hs, hprev, gates, u, ms)
            else:
                K.dense_fwd(feat, self.p("head.Ws"), self.p("head.bs"), u)
                K.dense_fwd(u, self.p("head.Wms"), self.p("head.bms"), ms)
            act["u"], act["ms"] = u, ms
        else:
            v = A.get(f"{tag}.v", Rh, self.n_out)
            K.dense_fwd(feat, self.p("head.Wo"), self.p("head.bo"), v)
            act["v"] = v
        return act

    # ---- backward --------------------------------------------------------------------------------------------------
    def backward(self, act, dout: torch.Tensor, tag: str = "b"):
        """dout = d loss / d ms [Rh,4] (policy) or d loss / d v [Rh,n_out] (values).  Accumulates into self.grads."""
        A = self.arena
        # the ~12 weight gradients of this pass defer their slab reductions to ONE batched launch (ops_nn.BwdWBatch); the
        # workspace is an arena buffer sized by the first pass (a later move bumps the arena generation)
        ws_floats = getattr(self, "_bwdw_floats", 16 << 20)

        def _alloc(n_floats):
            self._bwdw_floats = max(ws_floats, n_floats)
            return self.ws_arena.get(f"{tag}.bwdw_ws", self._bwdw_floats)
        batch = K.BwdWBatch(self.device, _alloc) if (dout.is_cuda and os.environ.get("DGPPO_NO_BWDW_BATCH") is None) else None
        if batch is not None:
            batch.__enter__()
        try:
            self._backward_body(act, dout, tag)
        except BaseException:
            if batch is not None:
                batch.__exit__(RuntimeError, None, None)
            raise
        if batch is not None:
            batch.__exit__(None, None, None)            # flush: the prepared-weight gradients are read below
        for l, (f, fp, d, kp) in enumerate(self.dims):
            q = lambda nm: self.p(f"gnn{l}.{nm}")
            gq = lambda nm: self.g(f"gnn{l}.{nm}")
            K.gnn_unprep(self.pg(f"gnn{l}.Mcat"), self.pg(f"gnn{l}.cvec"), self.pg(f"gnn{l}.Wout"), q("Wq"), q("bq"), q("Wk"),
                         gq("Wq"), gq("bq"), gq("Wk"), gq("Wv"), gq("bv"), gq("We"), gq("Wu"), f, fp, d, H_HEADS, kp)
        self.prep_grads.zero_()

    def _backward_body(self, act, dout: torch.Tensor, tag: str):
        cfg, A = self.cfg, self.arena
        G, Rh, n_inner, n_seq, T = act["G"], act["Rh"], act["n_inner"], act["n_seq"], act["T"]
        n = cfg.n_agents
        R = G * n
        feats: GraphFeats = act["feats"]
        Ro = G * feats.n_other
        feat = act["feat"]
        # the MLP trunk + GRU input projection ran as the fused forward kernel (64-wide chain, one GRU layer): its backward chain
        # is fused as well (DGPPO_NO_FUSED_TRUNK_BWD=1: the separate launches, for A/B runs and as the tests' second opinion)
        fused_trunk = (self.rnn == "gru" and self.rnn_layers == 1 and self.kind != "Vhg" and dout.is_cuda
                       and os.environ.get("DGPPO_NO_FUSED_TRUNK_BWD") is None)
        dgi0 = None
        dhs = A.get(f"{tag}.dhs", Rh, HID)
        if self.kind == "policy":
            K.dense_bwd_w(act["u"], dout, self.g("head.Wms"), self.g("head.bms"))
            du = A.get(f"{tag}.du", Rh, HID)
            K.dense_fwd(dout, self.p("head.Wms"), None, du, trans_w=True)
            K.dense_bwd_w(feat, du, self.g("head.Ws"), self.g("head.bs"))
            K.dense_fwd(du, self.p("head.Ws"), None, dhs, trans_w=True)
        else:
            K.dense_bwd_w(feat, dout, self.g("head.Wo"), self.g("head.bo"))
            K.dense_fwd(dout, self.p("head.Wo"), None, dhs, trans_w=True)
        if self.rnn == "none":
            dy = dhs                                   # the head reads the MLP output directly
        elif self.rnn == "lstm":
            stack = act["stack"]
            for l in range(len(stack) - 1, -1, -1):
                st = stack[l]
                dz = A.get(f"{tag}.dz_lstm{l}", Rh, 4 * HID)
                K.lstm_bwd(dhs, self.p(f"lstm{l}.Wh"), st["cprev"], st["gates"], dz, n_seq, T, n_inner)
                for half in (slice(0, 2 * HID), slice(2 * HID, 4 * HID)):
                    K.dense_bwd_w(st["hprev"], dz[:, half], self.g(f"lstm{l}.Wh")[:, half], self.g(f"lstm{l}.bh")[half])
                    K.dense_bwd_w(st["x"], dz[:, half], self.g(f"lstm{l}.Wi")[:, half], None)
                dx = A.get(f"{tag}.dy" if l == 0 else f"{tag}.dx{l}", Rh, HID)
                K.dense_fwd(dz, self.p(f"lstm{l}.Wi"), None, dx, trans_w=True)
                dhs = dx
            dy = dhs
        else:
            stack = act.get("stack") or [dict(x=None, gi=act["gi"], hs=act["hs"], hprev=act["hprev"], gates=act["gates"])]
            for l in range(len(stack) - 1, -1, -1):
                st = stack[l]
                pre = "gru" if l == 0 else f"gru{l}"
                dgi = A.get(f"{tag}.dgi{l}", Rh, 3 * HID)
                dgh = A.get(f"{tag}.dgh{l}", Rh, 3 * HID)
                K.gru_bwd(dhs, self.p(f"{pre}.Wh"), st["hprev"], st["gates"], dgi, dgh, n_seq, T, n_inner)
                # hr / hz have no bias (flax GRUCell); only the hn column block carries one
                K.dense_bwd_w(st["hprev"], dgh[:, :2 * HID], self.g(f"{pre}.Wh")[:, :2 * HID], None)
                K.dense_bwd_w(st["hprev"], dgh[:, 2 * HID:], self.g(f"{pre}.Wh")[:, 2 * HID:], self.g(f"{pre}.bhn"))
                x_l = act["y2"] if l == 0 else st["x"]
                K.dense_bwd_w(x_l, dgi, self.g(f"{pre}.Wi"), self.g(f"{pre}.bi"))
                if l == 0 and fused_trunk:
                    dgi0 = dgi                         # the fused trunk backward below starts from the gate-input gradient
                    break
                dx = A.get(f"{tag}.dy" if l == 0 else f"{tag}.dx{l}", Rh, HID)
                K.dense_fwd(dgi, self.p(f"{pre}.Wi"), None, dx, trans_w=True)
                dhs = dx                               # gradient of the layer below's output sequence
            dy = dhs
        x_in = {1: act["mlp_in"], 2: act["y1"]}
        top = act[f"Xa{self.gnn_layers}"]          # output of the last GNN layer (a ReLU output): masks the gradient entering it
        if fused_trunk:
            # dgi -> (Wi^T) -> LN+ReLU' -> (W2^T) -> LN+ReLU' -> (W1^T) -> dx in ONE launch (dgppo_mlp_gi_bwd): five launches and two
            # [Rh, 64] round trips through HBM less per network; the two weight gradients read the dpre buffers it writes
            dpre2, dpre1 = A.get(f"{tag}.dpre2", Rh, HID), A.get(f"{tag}.dpre1", Rh, HID)
            dy = A.get(f"{tag}.dyy1", Rh, HID)
            K.mlp_gi_bwd(dgi0, self.p("gru.Wi"), self.p("mlp.W2"), self.p("mlp.W1"), self.p("mlp.g2"), self.p("mlp.g1"),
                         act["p2"], act["y2"], act["st2"], act["p1"], act["y1"], act["st1"],
                         top if self.kind in ("policy", "Vh") else None, dpre2, dpre1, dy,
                         self.g("mlp.g2"), self.g("mlp.be2"), self.g("mlp.g1"), self.g("mlp.be1"))
            K.dense_bwd_w(x_in[2], dpre2, self.g("mlp.W2"), self.g("mlp.b2"))
            K.dense_bwd_w(x_in[1], dpre1, self.g("mlp.W1"), self.g("mlp.b1"))
        for i in (() if fused_trunk else (2, 1)):
            dpre = A.get(f"{tag}.dpre{i}", Rh, HID)
            K.ln_relu_bwd(act[f"p{i}"], act[f"y{i}"], act[f"st{i}"], self.p(f"mlp.g{i}"), dy, dpre, self.g(f"mlp.g{i}"),
                          self.g(f"mlp.be{i}"))
            K.dense_bwd_w(x_in[i], dpre, self.g(f"mlp.W{i}"), self.g(f"mlp.b{i}"))
            dy = A.get(f"{tag}.dyy{i}", Rh, HID if not (self.kind == "Vhg" and i == 1) else 2 * HID)
            # i == 1 for the per-agent nets: this IS the gradient of the last GNN layer's output -> ReLU backward fused here
            K.dense_fwd(dpre, self.p(f"mlp.W{i}"), None, dy, trans_w=True,
                        relu_mask=top if (i == 1 and self.kind in ("policy", "Vh")) else None)
        if self.kind == "Vl":
            dXa = A.get(f"{tag}.dXaL", R, OUT_DIM)
            K.mean_agents(dy, dXa, G, n, OUT_DIM, backward=True, relu_mask=top)
        elif self.kind == "Vhg":
            # dy [R, 128] = gradient of [x_i | tiled mean]: d x_i = dy[:, :64] + (1/n) sum_j dy[g, j, 64:]  (then relu')
            dXa = A.get(f"{tag}.dXaL", R, OUT_DIM)
            dXa.copy_(dy[:, :OUT_DIM])
            dglob = A.get(f"{tag}.dglob", R, OUT_DIM)
            dglob.copy_(dy[:, OUT_DIM:])
            dpool = A.get(f"{tag}.dpool", G, OUT_DIM)
            K.mean_agents(dglob, dpool, G, n, OUT_DIM)                               # (1/n) sum over the graph's agents
            K.mean_agents(dpool, dXa, G, n, OUT_DIM, backward=2, relu_mask=top)      # broadcast-add, through relu'
        else:
            dXa = dy
        # From here on dXa / dXo arrive ALREADY multiplied by relu'(layer output): the kernel that finishes each gradient
        # applies the mask (dense_fwd / mean_agents / attn_bwd epilogues) instead of a separate elementwise pass.
        dXo = None
        for l in range(self.gnn_layers - 1, -1, -1):
            f, fp, d, kp = self.dims[l]
            K.dense_bwd_w(act[f"zcat{l}"], dXa, self.pg(f"gnn{l}.Wout"), self.g(f"gnn{l}.bu"))
            dz = A.get(f"{tag}.dz{l}", R, kp)
            K.dense_fwd(dXa, self.pp(f"gnn{l}.Wout"), None, dz, trans_w=True)
            if dXo is not None:  # other nodes of layer l+1: relu(W_u x + b_u)
                K.dense_bwd_w(act[f"Xo{l}"], dXo, self.pg(f"gnn{l}.Wout")[:fp], self.g(f"gnn{l}.bu"))
            dqt = A.get(f"{tag}.dqt{l}", R, H_HEADS * fp)
            need_dx = l > 0
            dXa_l = A.get(f"{tag}.dXa{l}", R, fp) if need_dx else None
            dXo_prev = dXo
            dXo_l = A.get(f"{tag}.dXo{l}", Ro, fp) if (need_dx and Ro > 0) else None
            more_dXo = dXo_prev is not None and dXo_l is not None       # a dense term still accumulates into dXo_l (>= 3 layers)
            xf = act.get(f"xo_fused{l}")
            if xf is not None and dXa_l is not None and not more_dXo and not os.environ.get("DGPPO_ATTN_XO_NO_DW"):
                # the gradient of the recomputed rows never leaves the kernel: it goes straight into dWout_{l-1}[:8] / dbu_{l-1}
                ws = A.get(f"{tag}.xo_ws", K.attn_xo_workspace_floats(G), 1).view(-1)
                K.attn_bwd_xo_dw(cfg, fp, H_HEADS, kp, dz, act[f"attn{l}"], act[f"qt{l}"], act[f"Xa{l}"], xf[0], xf[1], xf[2],
                                 feats.efeat, dqt, dXa_l, self.pg(f"gnn{l - 1}.Wout")[:self.dims[l - 1][1]], self.g(f"gnn{l - 1}.bu"), ws, G)
                dXo_l = None
            elif xf is not None:
                K.attn_bwd_xo(cfg, fp, H_HEADS, kp, dz, act[f"attn{l}"], act[f"qt{l}"], act[f"Xa{l}"], xf[0], xf[1], xf[2],
                              feats.efeat, dqt, dXa_l, dXo_l, G, relu_xo=(dXo_l is not None and not more_dXo))
            else:
                K.attn_bwd(cfg, fp, H_HEADS, kp, dz, act[f"attn{l}"], act[f"qt{l}"], act[f"Xa{l}"],
                           act[f"Xo{l}"] if Ro > 0 else None, feats.efeat, dqt, dXa_l, dXo_l, G,
                           relu_xo=(dXo_l is not None and not more_dXo))
            K.dense_bwd_w(act[f"Xa{l}"], dqt, self.pg(f"gnn{l}.Mcat"), self.pg(f"gnn{l}.cvec"))
            if need_dx:
                K.dense_fwd(dqt, self.pp(f"gnn{l}.Mcat"), None, dXa_l, accumulate=True, trans_w=True, relu_mask=act[f"Xa{l}"])
                if more_dXo:
                    K.dense_fwd(dXo_prev, self.pp(f"gnn{l}.Wout")[:fp], None, dXo_l, accumulate=True, trans_w=True,
                                relu_mask=act[f"Xo{l}"])
            dXa, dXo = dXa_l, dXo_l

    # ---- flax-tree interop ----------------------------------------------------------------------------
    def load_tree(self, tree: dict):
        """tree = {'params': {...}} with the flax names of SURVEY A.9 (numpy or torch leaves)."""
        t = tree["params"]
        to = lambda x: torch.as_tensor(np.asarray(x), dtype=torch.float32).to(self.device)
        if self.kind == "policy":
            base, gnn, head_name = t["PolicyNet_0"], t["PolicyNet_0"]["GraphTransformerGNN_0"], "PolicyGNNHead"
        else:
            base, gnn, head_name = t, t["GraphTransformerGNN_0"], "ValueGNNHead"
        for l in range(self.gnn_layers):
            gl = gnn[f"GraphTransformer_{l}"]
            for nm, dn, key in (("Wq", "Dense_0", "kernel"), ("bq", "Dense_0", "bias"), ("Wk", "Dense_1", "kernel"),
                                ("bk", "Dense_1", "bias"), ("Wv", "Dense_2", "kernel"), ("bv", "Dense_2", "bias"),
                                ("We", "Dense_3", "kernel"), ("Wu", "Dense_4", "kernel"), ("bu", "Dense_4", "bias")):
                self.p(f"gnn{l}.{nm}").copy_(to(gl[dn][key]))
        hd = base[head_name]
        for i in (1, 2):
            self.p(f"mlp.W{i}").copy_(to(hd[f"Dense_{i - 1}"]["kernel"]))
            self.p(f"mlp.b{i}").copy_(to(hd[f"Dense_{i - 1}"]["bias"]))
            self.p(f"mlp.g{i}").copy_(to(hd[f"LayerNorm_{i - 1}"]["scale"]))
            self.p(f"mlp.be{i}").copy_(to(hd[f"LayerNorm_{i - 1}"]["bias"]))
        # flax auto-names: each layer of RNN.__call__ instantiates the cell class once for the isinstance
        # probe and once for use (rnn.py:19-20), so layer l's parameters live in GRUCell_{2l+1}
        for l in range(self.rnn_layers if self.rnn == "lstm" else 0):
            # LSTMCell_{3l+2}: two isinstance probes (GRUCell, LSTMCell) precede the instance in use (rnn.py:19-23)
            lc = base["RNN_0"][f"LSTMCell_{3 * l + 2}"]
            self.p(f"lstm{l}.Wi").copy_(torch.cat([to(lc[k]["kernel"]) for k in ("ii", "if", "ig", "io")], dim=1))
            self.p(f"lstm{l}.Wh").copy_(torch.cat([to(lc[k]["kernel"]) for k in ("hi", "hf", "hg", "ho")], dim=1))
            self.p(f"lstm{l}.bh").copy_(torch.cat([to(lc[k]["bias"]) for k in ("hi", "hf", "hg", "ho")]))
        for l in range(self.rnn_layers if self.rnn == "gru" else 0):
            pre = "gru" if l == 0 else f"gru{l}"
            gr = base["RNN_0"][f"GRUCell_{2 * l + 1}"]
            self.p(f"{pre}.Wi").copy_(torch.cat([to(gr[k]["kernel"]) for k in ("ir", "iz", "in")], dim=1))
            self.p(f"{pre}.bi").copy_(torch.cat([to(gr[k]["bias"]) for k in ("ir", "iz", "in")]))
            self.p(f"{pre}.Wh").copy_(torch.cat([to(gr[k]["kernel"]) for k in ("hr", "hz", "hn")], dim=1))
            self.p(f"{pre}.bhn").copy_(to(gr["hn"]["bias"]))
        if self.kind == "policy":
            self.p("head.Ws").copy_(to(t["ScaleHid"]["kernel"]))
            self.p("head.bs").copy_(to(t["ScaleHid"]["bias"]))
            self.p("head.Wms").copy_(torch.cat([to(t["OutputDenseMean"]["kernel"]), to(t["OutputDenseStdTrans"]["kernel"])], 1))
            self.p("head.bms").copy_(torch.cat([to(t["OutputDenseMean"]["bias"]), to(t["OutputDenseStdTrans"]["bias"])]))
        else:
            self.p("head.Wo").copy_(to(t["Dense_0"]["kernel"]))
            self.p("head.bo").copy_(to(t["Dense_0"]["bias"]))
        self.prepare()

    def to_tree(self, flat: Optional[torch.Tensor] = None) -> dict:
        """flat buffer (params by default, or grads) -> nested dict of numpy arrays with the flax names."""
        flat = self.params if flat is None else flat
        v = lambda nm: self.layout.view(flat, nm).detach().cpu().numpy().copy()
        gnn = {}
        for l in range(self.gnn_layers):
            gnn[f"GraphTransformer_{l}"] = {
                "Dense_0": {"kernel": v(f"gnn{l}.Wq"), "bias": v(f"gnn{l}.bq")},
                "Dense_1": {"kernel": v(f"gnn{l}.Wk"), "bias": v(f"gnn{l}.bk")},
                "Dense_2": {"kernel": v(f"gnn{l}.Wv"), "bias": v(f"gnn{l}.bv")},
                "Dense_3": {"kernel": v(f"gnn{l}.We")},
                "Dense_4": {"kernel": v(f"gnn{l}.Wu"), "bias": v(f"gnn{l}.bu")},
            }
        head = {}
        for i in (1, 2):
            head[f"Dense_{i - 1}"] = {"kernel": v(f"mlp.W{i}"), "bias": v(f"mlp.b{i}")}
            head[f"LayerNorm_{i - 1}"] = {"scale": v(f"mlp.g{i}"), "bias": v(f"mlp.be{i}")}
        rnn = {}
        for l in range(self.rnn_layers if self.rnn == "lstm" else 0):
            Wi, Wh, bh = v(f"lstm{l}.Wi"), v(f"lstm{l}.Wh"), v(f"lstm{l}.bh")
            cell = {}
            for q, g_ in enumerate("ifgo"):
                cell["i" + g_] = {"kernel": Wi[:, q * 64:(q + 1) * 64]}
                cell["h" + g_] = {"kernel": Wh[:, q * 64:(q + 1) * 64], "bias": bh[q * 64:(q + 1) * 64]}
            rnn[f"LSTMCell_{3 * l + 2}"] = cell
        for l in range(self.rnn_layers if self.rnn == "gru" else 0):
            pre = "gru" if l == 0 else f"gru{l}"
            Wi, bi, Wh = v(f"{pre}.Wi"), v(f"{pre}.bi"), v(f"{pre}.Wh")
            rnn[f"GRUCell_{2 * l + 1}"] = {
                "ir": {"kernel": Wi[:, :64], "bias": bi[:64]}, "iz": {"kernel": Wi[:, 64:128], "bias": bi[64:128]},
                "in": {"kernel": Wi[:, 128:], "bias": bi[128:]}, "hr": {"kernel": Wh[:, :64]}, "hz": {"kernel": Wh[:, 64:128]},
                "hn": {"kernel": Wh[:, 128:], "bias": v(f"{pre}.bhn")}}
        if self.kind == "policy":
            Wms, bms = v("head.Wms"), v("head.bms")
            k = self.n_out
            return {"params": {
                "PolicyNet_0": dict({"GraphTransformerGNN_0": gnn, "PolicyGNNHead": head}, **({"RNN_0": rnn} if rnn else {})),
                "ScaleHid": {"kernel": v("head.Ws"), "bias": v("head.bs")},
                "OutputDenseMean": {"kernel": Wms[:, :k], "bias": bms[:k]},
                "OutputDenseStdTrans": {"kernel": Wms[:, k:], "bias": bms[k:]}}}
        return {"params": dict({"GraphTransformerGNN_0": gnn, "ValueGNNHead": head, "Dense_0": {"kernel": v("head.Wo"), "bias": v("head.bo")}},
                               **({"RNN_0": rnn} if rnn else {}))}
