"""Batched DGPPO engine: rollout collection and the update, orchestrating the HIP kernels (no torch arithmetic on the
hot path — torch tensors are storage, copies/transposes are data movement).

Reference behaviour reproduced (file:line relative to /root/reference):
  rollout / test_rollout            dgppo/trainer/utils.py:22-86   (pre-step carry stored when stochastic, post-step when det)
  DGPPO.update / update_inner       dgppo/algo/dgppo.py:136-294
  update_Vl / update_policy         dgppo/algo/informarl.py:357-457
  update_Vh                         dgppo/algo/dgppo.py:296-321
"""
from __future__ import annotations

import dataclasses
from typing import Callable, Dict, Optional

import numpy as np
import os
import torch

from . import _native as N
from . import nets
from . import ops_algo as OA
from . import ops_env as OE
from . import ops_nn as K


@dataclasses.dataclass
class Hyper:
    gamma: float = 0.99
    gae_lambda: float = 0.95
    clip_eps: float = 0.25
    coef_ent: float = 1e-2
    max_grad_norm: float = 2.0
    lr_actor: float = 3e-4
    lr_Vl: float = 1e-3
    lr_Vh: float = 1e-3
    batch_size: int = 16384
    rnn_step: int = 16
    alpha: float = 10.0
    cbf_eps: float = 1e-2
    cbf_weight: float = 1.0
    cbf_schedule: bool = True
    train_steps: int = 100000
    actor_gnn_layers: int = 2
    Vl_gnn_layers: int = 2
    Vh_gnn_layers: int = 1
    use_rnn: bool = True              # --no-rnn: networks without a recurrent cell (policy.py:31-32, value.py:38-39)
    use_lstm: bool = False            # --use-lstm: flax LSTMCell instead of GRUCell in the policy and Vl (rnn.py:22-24)
    rnn_layers: int = 1               # stacked cells (dgppo/nn/rnn.py:17-29); the DGPPO constraint-value net keeps ONE
    lagr_init: float = 0.78           # InforMARL-Lagrangian: initial multipliers, their step size (informarl_lagr.py:52-53)
    lr_lagr: float = 1e-7
    cost_weight: float = 0.0          # InforMARL: weight of sum(max(cost, 0)) in the stage cost (informarl.py:329)
    cost_schedule: bool = False       # InforMARL: x5 at 50 % and again x5 at 75 % of train_steps (informarl.py:189-198)


class RolloutData:
    """Compact rollout record (SURVEY §7 'compact rollout storage'): time-major while collecting, env-major afterwards."""

    def __init__(self, cfg: N.EnvCfg, B: int, T: int, device, stochastic: bool, carry_dim: int = nets.HID):
        n, sd = cfg.n_agents, cfg.state_dim
        self.cfg, self.B, self.T, self.stochastic = cfg, B, T, stochastic
        self.has_hits = cfg.is_lidar and cfg.n_obs > 0
        z = lambda *s: torch.empty(*s, device=device)
        self.agent_tm = z(T + 1, B, n, sd)
        self.hits_tm = z(T + 1, B, n, cfg.top_k, 2) if self.has_hits else None
        self.goal = z(B, cfg.n_goals, sd)
        self.obst = z(B, cfg.n_obs, cfg.obst_stride) if cfg.n_obs > 0 else None
        self.action_tm = z(T, B, n, 2)
        self.log_pi_tm = z(T, B, n) if stochastic else None
        self.rnn_tm = z(T + 1, B, n, carry_dim)          # packed actor carry [h_0 | h_1 | ...] of the stacked cells
        self.reward_tm = z(T, B)
        self.cost_tm = z(T, B, n, cfg.n_cost)
        self._env_major = False

    def finalize(self):
        """time-major -> env-major copies (pure data movement).  The env-major buffers are allocated once per record and
        rewritten in place afterwards, so that device pointers stay valid for captured HIP graphs of the update."""
        if self._env_major:
            return self

        def tr(name, x):
            if x is None:
                return None
            cur = getattr(self, name, None)
            want = (x.shape[1], x.shape[0]) + tuple(x.shape[2:])
            if cur is None or tuple(cur.shape) != want:
                cur = torch.empty(want, device=x.device, dtype=x.dtype)
            cur.copy_(x.transpose(0, 1))
            return cur
        self.agent = tr("agent", self.agent_tm)                       # [B, T+1, n, sd]
        self.hits = tr("hits", self.hits_tm) if self.has_hits else None
        self.actions = tr("actions", self.action_tm)                  # [B, T, n, 2]
        self.log_pis = tr("log_pis", self.log_pi_tm) if self.stochastic else None
        self._rnn_em = tr("_rnn_em", self.rnn_tm)                     # [B, T+1, n, 64]
        # stored carry of step t: pre-step (rollout, trainer/utils.py:46-51) or post-step (test_rollout, :71-77)
        self.rnn_states = self._rnn_em[:, :self.T] if self.stochastic else self._rnn_em[:, 1:]
        self.rewards = tr("rewards", self.reward_tm)                  # [B, T]
        self.costs = tr("costs", self.cost_tm)                        # [B, T, n, n_cost]
        self._env_major = True
        return self


class OptState:
    def __init__(self, n: int, device):
        self.m = torch.zeros(n, device=device)
        self.v = torch.zeros(n, device=device)
        self.state = torch.zeros(N.OPT_STATE_FLOATS, device=device)


class Engine:
    def __init__(self, cfg: N.EnvCfg, hyper: Hyper, device, T: int = 128,
                 allreduce: Optional[Callable[[torch.Tensor], None]] = None, world: int = 1, prepass_graphs: int = 1 << 20,
                 use_graphs: bool = False, multi_stream: bool = False, algo: str = "dgppo", rank: int = 0):
        self.cfg, self.hp, self.device, self.T = cfg, hyper, device, T
        assert algo in ("dgppo", "informarl", "hcbfcrpo", "informarl_lagr"), algo
        # "informarl" / "hcbfcrpo": no constraint-value network and no deterministic rollout (informarl.py, hcbfcrpo.py);
        # "informarl_lagr": a constraint-value network with global information and its own recurrent carry, trained on the
        # stochastic rollout, plus per-(agent, component) Lagrange multipliers (informarl_lagr.py)
        self.algo = algo
        # HIP-graph replay of the launch-bound rollout loop (18 small kernels per env step).  Opt-in because the record
        # buffers then belong to the engine: a RolloutData stays valid only until the next rollout of the same kind.
        self.use_graphs = use_graphs and os.environ.get("DGPPO_HIPGRAPH", "1") != "0"
        # run the Vl / Vh / policy updates of a minibatch on three HIP streams (they are independent)
        self.multi_stream = multi_stream and os.environ.get("DGPPO_MULTI_STREAM", "1") != "0"
        self._side_streams = None
        self._ro_cache: Dict[tuple, dict] = {}
        self._upd_graph: dict = {}
        self.n_cost = cfg.n_cost              # 2, or 3 with MPEConnectSpread's connectivity cost
        # ONE flat fp32 buffer [g_policy | g_Vl | g_Vh | scalars]: each network's gradient buffer is a
        # 16-byte-aligned slice of it and the loss/metric sums of the minibatch (stats rows 0..2) sit at its tail, so the
        # data-parallel update needs a single all-reduce per minibatch.  Row 3 of the stats (the per-iteration safe count)
        # lies outside the reduced range.
        spec = [("policy", "policy", hyper.actor_gnn_layers, 2), ("Vl", "Vl", hyper.Vl_gnn_layers, 1)]
        if algo == "dgppo":
            spec.append(("Vh", "Vh", hyper.Vh_gnn_layers, self.n_cost))
        elif algo == "informarl_lagr":
            spec.append(("Vh", "Vhg", hyper.Vh_gnn_layers, self.n_cost))
        # recurrent options (train.py --no-rnn / --rnn-layers): the policy and Vl (and the Lagrangian Vh) stack rnn_layers
        # cells; DGPPO's constraint-value net is built with the ValueNet default of ONE layer (dgppo.py:83-95) and reads layer
        # 0 of the actor's carry
        def rnn_kw(name):
            if not hyper.use_rnn:
                return dict(rnn="none", rnn_layers=0)
            if name == "Vh" and algo == "dgppo":        # ValueNet(use_lstm=False) with the default single cell (dgppo.py:83-95)
                return dict(rnn="gru", rnn_layers=1)
            return dict(rnn="lstm" if hyper.use_lstm else "gru", rnn_layers=hyper.rnn_layers)
        sizes = [nets.make_layout(kind, cfg.node_dim, layers, n_out, **rnn_kw(k)).size for k, kind, layers, n_out in spec]
        offs, tot = [], 0
        for sz in sizes:
            offs.append(tot)
            tot += (sz + 3) // 4 * 4
        self.flat_grads = torch.zeros(tot + 4 * 8, device=device)
        self.n_reduced = tot + 3 * 8
        self.stats = self.flat_grads[tot:tot + 32].view(4, 8)
        built = {k: nets.Net(kind, cfg, layers, n_out, device, grads=self.flat_grads[o:o + sz], **rnn_kw(k))
                 for (k, kind, layers, n_out), o, sz in zip(spec, offs, sizes)}
        if algo == "informarl_lagr":
            self.lagr = torch.full((cfg.n_agents, self.n_cost), float(hyper.lagr_init), device=device)   # informarl_lagr.py:107
            self.lagr_sums = torch.zeros(cfg.n_agents * self.n_cost, device=device)
        self.policy, self.Vl, self.Vh = built["policy"], built["Vl"], built.get("Vh")
        self.HC = self.policy.carry_dim              # width of the stored actor carry
        self.opt = {k: OptState(net.layout.size, device) for k, net in self.nets.items()}
        self.arena = nets.Arena(device)
        # allreduce(flat): in-place SUM over the data-parallel ranks of the flat buffer above (dgppo_comm_allreduce_sum_f32 on
        # the caller's stream); the 1/world is applied inside dgppo_clip_adam_step (grad_scale), not by a separate pass
        self.allreduce = allreduce
        self.world, self.rank = int(world), int(rank)
        assert self.world >= 1 and (allreduce is not None or self.world == 1), "world > 1 needs an allreduce"
        assert 0 <= self.rank < self.world, f"rank {rank} outside world {world}"
        self.prepass_graphs = int(os.environ.get("DGPPO_PREPASS_GRAPHS", prepass_graphs))      # tuning override
        self.ray_cos, self.ray_sin = (OE.ray_tables(cfg.n_rays, device) if cfg.is_lidar else (None, None))
        self.lam_pow = OA.lam_pow_table(hyper.gae_lambda, T, device)
        # the constant entropy noise of SURVEY A.7 (distribution.py:40: seed drawn once at trace time)
        self.eps_hat = torch.zeros(cfg.n_agents, 2, device=device)
        # envs whose reset could not place a valid scene within the kernels' loop bounds (dgppo_env_reset_checked): counted on
        # the device by every rollout, read at the iteration's one host sync (info) — training on such a batch is an error
        self.reset_failed = torch.zeros(1, dtype=torch.int32, device=device)
        self.grad_hook: Optional[Callable[[str, nets.Net, int], None]] = None   # (net name, net, minibatch) before the optimiser
        self._mb = 0

    @property
    def nets(self) -> Dict[str, nets.Net]:
        d = {"policy": self.policy, "Vl": self.Vl}
        if self.Vh is not None:
            d["Vh"] = self.Vh
        return d

    def set_entropy_noise(self, seed: int):
        OE.randn(seed, 0, self.eps_hat.view(-1))

    # ------------------------------------------------------------------------------------------------------------------
    # rollout
    # ------------------------------------------------------------------------------------------------------------------
    def _feats_at(self, tag, agent_slab, hits_slab, goal, obst, B):
        """graph features of B dense [B, n, sd] states."""
        cfg = self.cfg
        f = nets.GraphFeats(cfg, B, self.arena, tag)
        n, sd = cfg.n_agents, cfg.state_dim
        f.compute(agent_slab, n * sd, 0, goal, obst, hits_slab, n * cfg.top_k * 2, 0, None, B, 1)
        return f

    def _rollout_steps(self, ro: RolloutData, eps, B: int, stochastic: bool):
        """the T env steps of a rollout: policy forward (GNN + GRU + head) and env.step, all on the current stream."""
        cfg, T, n = self.cfg, self.T, self.cfg.n_agents
        tag = "ro" if stochastic else "rod"    # separate scratch per kind: the two rollouts may run on different streams
        for t in range(T):
            hits_t = ro.hits_tm[t] if ro.has_hits else None
            feats = self._feats_at(tag, ro.agent_tm[t], hits_t, ro.goal, ro.obst, B)
            act = self.policy.forward(feats, n_seq=B * n, T=1, h0=ro.rnn_tm[t].view(B * n, self.HC), tag=tag,
                                      hs_out=ro.rnn_tm[t + 1].view(B * n, self.HC), train=False)
            a_t = ro.action_tm[t].view(B * n, 2)
            if stochastic:
                K.policy_head(act["ms"], eps[t], None, a_t, ro.log_pi_tm[t].view(B * n), None, n, 0)
            else:
                K.policy_head(act["ms"], None, None, a_t, None, None, n, 1)
            OE.env_step(cfg, ro.agent_tm[t], ro.action_tm[t], ro.goal, ro.obst, hits_t, self.ray_cos, self.ray_sin,
                        ro.agent_tm[t + 1], ro.hits_tm[t + 1] if ro.has_hits else None, ro.reward_tm[t], ro.cost_tm[t], None)

    def rollout(self, seeds: torch.Tensor, stochastic: bool, noise_seed: int = 0) -> RolloutData:
        cfg, T = self.cfg, self.T
        B = int(seeds.shape[0])
        n = cfg.n_agents
        slot = None
        if self.use_graphs:
            # persistent record buffers per (B, kind): eager + capture on the first call, replayed afterwards
            slot = self._ro_cache.setdefault((B, stochastic), {"ro": None, "graph": None, "gen": -1, "calls": 0})
            if slot["ro"] is None:
                slot["ro"] = RolloutData(cfg, B, T, self.device, stochastic, self.HC)
            ro = slot["ro"]
            ro._env_major = False
        else:
            ro = RolloutData(cfg, B, T, self.device, stochastic, self.HC)
        OE.env_reset(cfg, seeds, ro.agent_tm[0], ro.goal, ro.obst, self.reset_failed)
        if ro.has_hits:
            OE.env_step(cfg, ro.agent_tm[0], None, ro.goal, ro.obst, None, self.ray_cos, self.ray_sin, None, ro.hits_tm[0],
                        None, None, None)
        ro.rnn_tm[0].zero_()                                   # init_rnn_state = zeros (informarl.py:115-124)
        eps = None
        if stochastic:
            # the sampling noise of step t is one row of world * B * n * 2 normals, of which this rank fills the window of
            # its envs: the union of the ranks' rollouts is the single-device rollout of the global batch
            eps = self.arena.get("ro.eps", T, B * n, 2)
            OE.randn_rows(noise_seed, eps.view(T, B * n * 2), self.world * B * n * 2, self.rank * B * n * 2)
        if slot is None:
            self._rollout_steps(ro, eps, B, stochastic)
            return ro
        slot["calls"] += 1
        if slot["graph"] is not None and slot["gen"] != self._arena_generation():
            slot["graph"] = None                               # a scratch buffer moved: the captured pointers are stale
        if slot["graph"] is not None:
            slot["graph"].replay()
            K.FLOPS[0] += slot.get("flops", 0.0)               # the replayed launches never pass through the Python wrappers
            return ro
        # first call (or stale graph): eager launches — they also size every scratch buffer — then capture the same loop
        # right away, so that the second call already replays (one warm-up iteration is enough for steady state)
        f0 = K.FLOPS[0]
        self._rollout_steps(ro, eps, B, stochastic)
        slot["flops"] = K.FLOPS[0] - f0
        if not slot.get("failed", False):
            graph = torch.cuda.CUDAGraph()
            try:
                # thread-local error mode: other threads (the RCCL watchdog of torch.distributed) keep issuing HIP calls
                # while this thread captures; in the default global mode those would invalidate the capture.
                # Capturing records the launches without executing them: the results of the eager pass above stand.
                f1 = K.FLOPS[0]
                with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                    self._rollout_steps(ro, eps, B, stochastic)
                K.FLOPS[0] = f1 + slot["flops"]                 # capture records, the replay below executes once
                slot["graph"], slot["gen"] = graph, self._arena_generation()
                # the first launch of a ~2300-node graph uploads it to the device (~100 ms): pay that here, in the same
                # warm-up call (same inputs, so it rewrites the record with identical values)
                graph.replay()
            except Exception as ex:                             # keep training: eager launches are always correct
                slot["failed"] = True
                torch.cuda.synchronize()
                print(f"[dgppo_amd] HIP-graph capture of the rollout loop failed ({type(ex).__name__}: {ex}); "
                      f"continuing with eager launches", flush=True)
        return ro

    def _update_generation(self) -> int:
        """scratch buffers the minibatch step touches: the engine's arena and every network's"""
        return self.arena.generation + sum(net.arena.generation + net.ws_arena.generation for net in self.nets.values())

    def _arena_generation(self) -> int:
        """scratch buffers the rollout loop touches live in the engine's arena (features) and the policy's (activations)"""
        return self.arena.generation + self.policy.arena.generation

    def rollout_pair(self, seeds: torch.Tensor, det_seeds: torch.Tensor, noise_seed: int = 0):
        """the stochastic training rollout and the deterministic rollout of the same parameters (informarl.py:254-256 and
        dgppo.py:139-141).  They are independent, so with multi_stream they run on two HIP streams side by side."""
        if not (self.multi_stream and self.device.type == "cuda"):
            return self.rollout(seeds, True, noise_seed), self.rollout(det_seeds, False)
        main = torch.cuda.current_stream(self.device)
        s0, s1 = self._net_streams()[:2]
        out = [None, None]
        for k, (st, sd, stoch) in enumerate(((s0, seeds, True), (s1, det_seeds, False))):
            st.wait_stream(main)
            with torch.cuda.stream(st):
                out[k] = self.rollout(sd, stoch, noise_seed if stoch else 0)
        main.wait_stream(s0)
        main.wait_stream(s1)
        return out[0], out[1]

    # ------------------------------------------------------------------------------------------------------------------
    # value pre-passes (dgppo.py:204-229, 262-264)
    # ------------------------------------------------------------------------------------------------------------------
    def _block_feats(self, tag, ro: RolloutData, e0, Eb, t0, n_time, env_ids=None):
        cfg = self.cfg
        n, sd, T1 = cfg.n_agents, cfg.state_dim, self.T + 1
        f = nets.GraphFeats(cfg, Eb * n_time, self.arena, tag)
        agent = ro.agent[e0:, t0] if env_ids is None else ro.agent[:, t0]
        hits = None
        if ro.has_hits:
            hits = ro.hits[e0:, t0] if env_ids is None else ro.hits[:, t0]
        goal = ro.goal[e0:] if env_ids is None else ro.goal
        obst = None
        if ro.obst is not None:
            obst = ro.obst[e0:] if env_ids is None else ro.obst
        f.compute(agent, T1 * n * sd, n * sd, goal, obst, hits, T1 * n * cfg.top_k * 2, n * cfg.top_k * 2, env_ids, Eb, n_time)
        return f

    def values_prepass(self, ro: RolloutData, want_Vl: bool, want_Vh: bool = True):
        """-> Vl [B,T+1] (or None), Vh [B,T+1,n,nh] of one rollout, with the reference's carry conventions."""
        cfg, T, B = self.cfg, self.T, ro.B
        n, nh, H = cfg.n_agents, self.n_cost, nets.HID
        kind = "ro" if ro.stochastic else "det"      # persistent outputs (stable pointers for the captured update graph)
        Vl_buf = self.arena.get(f"tg.Vl.{kind}", B, T + 1) if want_Vl else None
        Vh_buf = self.arena.get(f"tg.Vh.{kind}", B, T + 1, n, nh) if want_Vh else None
        block = max(1, min(B, self.prepass_graphs // (T + 1)))
        for e0 in range(0, B, block):
            Eb = min(block, B - e0)
            feats = self._block_feats("pre", ro, e0, Eb, 0, T + 1)
            if want_Vl:
                act = self.Vl.forward(feats, n_seq=Eb, T=T + 1, h0=None, tag="pre", train=False)
                Vl_buf[e0:e0 + Eb].copy_(act["v"].view(Eb, T + 1))
            if not want_Vh:
                continue
            # final carry: actor GRU on next_graph[-1] from rnn_states[-1] (dgppo.py:222-226)
            fin = self._block_feats("fin", ro, e0, Eb, T, 1)
            HC = self.HC
            h_last = self.arena.get("pre.hlast", Eb * n, HC)
            h_last.view(Eb, n, HC).copy_(ro.rnn_states[e0:e0 + Eb, T - 1])
            h0_all = self.arena.get("pre.h0all", Eb, T + 1, n, H)
            hstar = self.arena.get("pre.hstar", Eb * n, HC)
            self.policy.forward(fin, n_seq=Eb * n, T=1, h0=h_last, tag="fin", hs_out=hstar, train=False)
            # the constraint-value net has ONE cell and reads layer 0 of the actor's packed carry (rnn.py:20: rnn_state[0])
            h0_all[:, :T].copy_(ro.rnn_states[e0:e0 + Eb][..., :H])
            h0_all[:, T].copy_(hstar.view(Eb, n, HC)[..., :H])
            act = self.Vh.forward(feats, n_seq=Eb * (T + 1) * n, T=1, h0=h0_all.view(-1, H) if self.hp.use_rnn else None,
                                  tag="pre", train=False)
            Vh_buf[e0:e0 + Eb].copy_(act["v"].view(Eb, T + 1, n, nh))
        return Vl_buf, Vh_buf

    # ------------------------------------------------------------------------------------------------------------------
    # update
    # ------------------------------------------------------------------------------------------------------------------
    def cbf_weight_at(self, step: int) -> float:
        hp = self.hp
        w = hp.cbf_weight
        if hp.cbf_schedule:  # optax.piecewise_constant_schedule (dgppo.py:73-80)
            if step >= int(hp.train_steps * 0.5):
                w *= 2
            if step >= int(hp.train_steps * 0.75):
                w *= 2
        return w

    def cost_weight_at(self, step: int) -> float:
        hp = self.hp
        w = hp.cost_weight
        if hp.cost_schedule:  # optax.piecewise_constant_schedule(init, {0.5*steps: 5, 0.75*steps: 5}) (informarl.py:189-196)
            if step >= int(hp.train_steps * 0.5):
                w *= 5
            if step >= int(hp.train_steps * 0.75):
                w *= 5
        return w

    def targets_hcbfcrpo(self, ro: RolloutData, step: int):
        """HCBFCRPO (hcbfcrpo.py:120-186): DGPPO's targets with the hand-crafted CBF Vh := env.get_cost(graph).  For
        t < T that is the stored cost (cost is evaluated on the pre-step graph); the final entry is the cost of
        next_graph[-1], obtained from one more env.step call whose other outputs are discarded."""
        cfg, T, B, hp = self.cfg, self.T, ro.B, self.hp
        n, nh, dev = cfg.n_agents, self.n_cost, self.device
        Vl, _ = self.values_prepass(ro, want_Vl=True, want_Vh=False)
        fin_agent = ro.agent[:, T].contiguous()
        fin_hits = ro.hits[:, T].contiguous() if ro.has_hits else None
        A = self.arena
        fin_cost = A.get("tg.fin_cost", B, n, nh)
        scratch_agent = torch.empty_like(fin_agent)
        OE.env_step(cfg, fin_agent, torch.zeros(B, n, 2, device=dev), ro.goal, ro.obst, fin_hits, self.ray_cos, self.ray_sin,
                    scratch_agent, torch.empty_like(fin_hits) if fin_hits is not None else None, torch.empty(B, device=dev),
                    fin_cost, None)
        Vh = A.get("tg.Vh.crafted", B, T + 1, n, nh)
        Vh[:, :T].copy_(ro.costs)
        Vh[:, T].copy_(fin_cost)
        Qh = A.get("tg.Qh", B, T, n, nh)
        Ql = A.get("tg.Ql", B, T)
        OA.gae(ro.costs, ro.rewards, Vh, Vl, self.lam_pow, hp.gamma, hp.gae_lambda, Qh, Ql)
        adv = A.get("tg.adv", B, T, n)
        self.stats.zero_()
        OA.advantage(Ql, Vl, Vh, cfg.dt, hp.alpha, hp.cbf_eps, self.cbf_weight_at(step), adv, self.stats[3])
        return dict(Vl=Vl, Vh=Vh, Ql=Ql, Qh=Qh, adv=adv)

    def targets_informarl(self, ro: RolloutData, step: int):
        """InforMARL (informarl.py:309-336): Vl pass, Dec-OCP GAE with Vh := Vl and the cost-shaped stage cost, advantage
        = -(Ql - Vl) normalised per env."""
        cfg, T, B, hp = self.cfg, self.T, ro.B, self.hp
        n, nh, dev = cfg.n_agents, self.n_cost, self.device
        Vl, _ = self.values_prepass(ro, want_Vl=True, want_Vh=False)
        A = self.arena
        Vh = A.get("tg.Vh.bcast", B, T + 1, n, nh)
        Vh.copy_(Vl.view(B, T + 1, 1, 1).expand(B, T + 1, n, nh))
        shaped = A.get("tg.shaped", B, T)
        OA.shaped_reward(ro.rewards, ro.costs, self.cost_weight_at(step), shaped)
        Qh = A.get("tg.Qh", B, T, n, nh)
        Ql = A.get("tg.Ql", B, T)
        OA.gae(ro.costs, shaped, Vh, Vl, self.lam_pow, hp.gamma, hp.gae_lambda, Qh, Ql)
        adv = A.get("tg.adv", B, T, n)
        self.stats.zero_()
        OA.advantage(Ql, Vl, None, cfg.dt, 0.0, 0.0, 0.0, adv, self.stats[3])
        return dict(Vl=Vl, Ql=Ql, Qh=Qh, adv=adv)

    def targets_lagr(self, ro: RolloutData, step: int):
        """InforMARL-Lagrangian (informarl_lagr.py:177-235): Vl and Vh scans (each with its own zero-initialised carry, final
        value on next_graph[-1] = one more scan step), Dec-OCP GAE on the clipped costs, advantage with the multipliers."""
        cfg, T, B, hp = self.cfg, self.T, ro.B, self.hp
        n, nh, A = cfg.n_agents, self.n_cost, self.arena
        Vl, _ = self.values_prepass(ro, want_Vl=True, want_Vh=False)
        Vh = A.get("tg.Vh.lagr", B, T + 1, n, nh)
        block = max(1, min(B, self.prepass_graphs // (T + 1)))
        for e0 in range(0, B, block):
            Eb = min(block, B - e0)
            feats = self._block_feats("pre", ro, e0, Eb, 0, T + 1)
            act = self.Vh.forward(feats, n_seq=Eb * n, T=T + 1, h0=None, tag="pre", train=False)
            Vh[e0:e0 + Eb].copy_(act["v"].view(Eb, T + 1, n, nh))
        cpos = A.get("tg.costs_pos", B, T, n, nh)
        OA.relu_fwd(ro.costs, cpos)                                  # jnp.clip(rollout.costs, a_min=0)
        Qh = A.get("tg.Qh", B, T, n, nh)
        Ql = A.get("tg.Ql", B, T)
        OA.gae(cpos, ro.rewards, Vh, Vl, self.lam_pow, hp.gamma, hp.gae_lambda, Qh, Ql)
        adv = A.get("tg.adv", B, T, n)
        Ah = A.get("tg.Ah", B, T, n, nh)
        self.stats.zero_()
        OA.advantage_lagr(Ql, Vl, Qh, Vh, self.lagr, adv, Ah)
        return dict(Vl=Vl, Vh=Vh, Ql=Ql, Qh=Qh, adv=adv, Ah=Ah)

    def _net_streams(self):
        if self._side_streams is None:
            self._side_streams = [torch.cuda.Stream(self.device) for _ in range(3)]
        return self._side_streams

    def _opt_step(self, name: str, lr: float):
        """NaN check -> norm -> clip -> Adam on one network (trainer/utils.py:89-118 + optax); in the data-parallel path the
        gradients were summed over the ranks just before and are read as g / world."""
        net, opt = self.nets[name], self.opt[name]
        if self.grad_hook is not None:
            self.grad_hook(name, net, self._mb)
        OA.clip_adam_step(net.params, net.grads, opt.m, opt.v, opt.state, lr, self.hp.max_grad_norm,
                          grad_scale=1.0 / self.world)
        net.prepare()

    def targets(self, ro: RolloutData, det: RolloutData, step: int):
        """Vl/Vh pre-passes, the two GAEs and the advantage merge (dgppo.py:204-273)."""
        cfg, T, B, hp = self.cfg, self.T, ro.B, self.hp
        n, nh = cfg.n_agents, self.n_cost
        Vl, Vh = self.values_prepass(ro, want_Vl=True)
        _, Vh_det = self.values_prepass(det, want_Vl=False)
        A = self.arena
        Qh = A.get("tg.Qh", B, T, n, nh)
        Ql = A.get("tg.Ql", B, T)
        OA.gae(ro.costs, ro.rewards, Vh, Vl, self.lam_pow, hp.gamma, hp.gae_lambda, Qh, Ql)
        Qh_det = A.get("tg.Qh_det", B, T, n, nh)
        Ql_det = A.get("tg.Ql_det", B, T)
        OA.gae(det.costs, det.rewards, Vh_det, Vl, self.lam_pow, hp.gamma, hp.gae_lambda, Qh_det, Ql_det)
        adv = A.get("tg.adv", B, T, n)
        self.stats.zero_()
        OA.advantage(Ql, Vl, Vh, cfg.dt, hp.alpha, hp.cbf_eps, self.cbf_weight_at(step), adv, self.stats[3])
        return dict(Vl=Vl, Vh=Vh, Vh_det=Vh_det, Ql=Ql, Qh=Qh, Qh_det=Qh_det, adv=adv)

    def update(self, ro: RolloutData, det: RolloutData, step: int, perm: np.ndarray) -> dict:
        cfg, T, B, hp = self.cfg, self.T, ro.B, self.hp
        n, nh, H = cfg.n_agents, self.n_cost, nets.HID
        informarl = self.algo in ("informarl", "hcbfcrpo")          # these baselines train only Vl and the policy
        lagr = self.algo == "informarl_lagr"
        import time as _time
        th = [_time.perf_counter()]                # host-side issue times of the phases (diagnostics: self.host_ms)
        ro.finalize()
        if self.algo == "dgppo":
            det.finalize()
        assert B * T >= hp.batch_size, "n_env_train * T must be >= batch_size (dgppo.py:153)"
        Eb = hp.batch_size // T
        assert B % Eb == 0 and T % hp.rnn_step == 0, "B % (batch_size // T) == 0 and T % rnn_step == 0 required (SURVEY A.11)"
        C = T // hp.rnn_step
        tg = (self.targets_informarl(ro, step) if self.algo == "informarl" else
              self.targets_hcbfcrpo(ro, step) if self.algo == "hcbfcrpo" else
              self.targets_lagr(ro, step) if lagr else self.targets(ro, det, step))
        th.append(_time.perf_counter())
        idx_all = torch.from_numpy(np.ascontiguousarray(perm.astype(np.int64))).to(self.device)
        n_mb = B // Eb
        G = Eb * T
        R = G * n
        is_cuda = self.device.type == "cuda"
        reduce = self.allreduce is not None
        A = self.arena
        # static inputs of one minibatch step: the env ids of the minibatch and everything gathered by them
        mb_idx = A.get("mb.idx", Eb, dtype=torch.int64)
        mb_idx32 = A.get("mb.idx32", Eb, dtype=torch.int32)
        Ql_mb = A.get("mb.Ql", Eb, T)
        act_mb = A.get("mb.act", Eb, T, n, 2)
        lp_old_mb = A.get("mb.lp_old", Eb, T, n)
        adv_mb = A.get("mb.adv", Eb, T, n)
        if self.algo == "dgppo":
            h0_det = A.get("mb.h0_det", Eb, T, n, H)
            Qh_det_mb = A.get("mb.Qh_det", Eb, T, n, nh)
        if lagr:
            Qh_mb = A.get("mb.Qh", Eb, T, n, nh)
            Vh_mb = A.get("mb.Vh", Eb, T + 1, n, nh)
            Ah_mb = A.get("mb.Ah", Eb, T, n, nh)

        ctx = {}

        def lagr_step(apply_now: bool):
            """update_lagr (informarl_lagr.py:286-309) with the UPDATED policy: log pi of the stored actions over whole
            episodes (zero carry), then the multiplier step — at once on one device; with a gradient exchange only this
            rank's sums here, the all-reduce of the sums and the step follow in exchange_lagr()."""
            full = self.policy.forward(ctx["feats"], n_seq=Eb * n, T=T, h0=None, tag="lg", train=False)
            lp_full = A.get("mb.lp_full", R)
            K.policy_head(full["ms"], self.eps_hat, act_mb.view(R, 2), None, lp_full, A.get("mb.ent_full", R), n, 2)
            if apply_now:
                OA.lagr_update(lp_full.view(Eb, T, n), lp_old_mb, Vh_mb, Ah_mb, self.lagr, self.lagr_sums, hp.gamma, hp.lr_lagr)
            else:
                OA.lagr_sums(lp_full.view(Eb, T, n), lp_old_mb, Vh_mb, Ah_mb, self.lagr_sums, hp.gamma)

        def body_pre():
            """device work of ONE minibatch up to the gradient exchange (dgppo.py:276-289): gathers and the three forward /
            backward passes (with the optimiser steps when there is no exchange) — reads the minibatch's env ids from
            mb_idx / mb_idx32."""
            main = torch.cuda.current_stream(self.device) if is_cuda else None
            side = self._net_streams() if (self.multi_stream and main is not None) else None
            self.stats[:3].zero_()
            # everything the three updates read is produced on the main stream first
            feats = ctx["feats"] = self._block_feats("mb", ro, 0, Eb, 0, T, env_ids=mb_idx32)
            torch.index_select(tg["Ql"], 0, mb_idx, out=Ql_mb)
            if self.algo == "dgppo":
                feats_det = self._block_feats("mbd", det, 0, Eb, 0, T, env_ids=mb_idx32)
                if self.HC == H:
                    torch.index_select(det.rnn_states, 0, mb_idx, out=h0_det)
                else:                                          # stacked cells: gather the packed carry, keep layer 0
                    full = A.get("mb.h0_det_full", Eb, T, n, self.HC)
                    torch.index_select(det.rnn_states, 0, mb_idx, out=full)
                    h0_det.copy_(full[..., :H])
                torch.index_select(tg["Qh_det"], 0, mb_idx, out=Qh_det_mb)
            if lagr:
                torch.index_select(tg["Qh"], 0, mb_idx, out=Qh_mb)
                torch.index_select(tg["Vh"], 0, mb_idx, out=Vh_mb)
                torch.index_select(tg["Ah"], 0, mb_idx, out=Ah_mb)
            torch.index_select(ro.actions, 0, mb_idx, out=act_mb)
            torch.index_select(ro.log_pis, 0, mb_idx, out=lp_old_mb)
            torch.index_select(tg["adv"], 0, mb_idx, out=adv_mb)

            def update_Vl():      # informarl.py:357-385: chunks of rnn_step with zero initial carry
                act = self.Vl.forward(feats, n_seq=Eb * C, T=hp.rnn_step, h0=None, tag="tr")
                dv = A.get("mb.dv", G, 1)
                K.value_loss(act["v"], Ql_mb.view(G, 1), dv, self.stats[0])
                self.Vl.zero_grads()
                self.Vl.backward(act, dv)
                if not reduce:
                    self._opt_step("Vl", hp.lr_Vl)

            def update_Vh():
                if lagr:          # informarl_lagr.py:252-284: chunks of rnn_step, zero initial carry, stochastic rollout
                    act = self.Vh.forward(feats, n_seq=Eb * C * n, T=hp.rnn_step, h0=None, tag="tr")
                    target = Qh_mb
                else:             # dgppo.py:296-321: the deterministic rollout with its stored carry
                    act = self.Vh.forward(feats_det, n_seq=R, T=1, h0=h0_det.view(R, H) if hp.use_rnn else None, tag="tr")
                    target = Qh_det_mb
                dvh = A.get("mb.dvh", R, nh)
                K.value_loss(act["v"], target.view(R, nh), dvh, self.stats[1])
                self.Vh.zero_grads()
                self.Vh.backward(act, dvh)
                if not reduce:
                    self._opt_step("Vh", hp.lr_Vh)

            def update_policy():  # informarl.py:405-457
                act = self.policy.forward(feats, n_seq=Eb * C * n, T=hp.rnn_step, h0=None, tag="tr")
                lp = A.get("mb.lp", R)
                ent = A.get("mb.ent", R)
                dms = A.get("mb.dms", R, 4)
                K.policy_head(act["ms"], self.eps_hat, act_mb.view(R, 2), None, lp, ent, n, 2, lp_old_mb.view(R),
                              adv_mb.view(R), dms, self.stats[2], hp.clip_eps, hp.coef_ent)
                self.policy.zero_grads()
                self.policy.backward(act, dms)
                if not reduce:
                    self._opt_step("policy", hp.lr_actor)
                if lagr and not reduce:
                    lagr_step(apply_now=True)

            if informarl:
                update_Vh = lambda: None                       # noqa: E731  (no constraint-value network)
            if side is None:
                update_Vl(); update_Vh(); update_policy()
            else:
                # the three networks share no state: one HIP stream each, joined before the next minibatch touches the
                # shared inputs again (most of these kernels are latency-bound and leave CUs idle on their own)
                ready = main.record_event()
                for fn, st in ((update_policy, side[0]), (update_Vl, side[1]), (update_Vh, side[2])):
                    with torch.cuda.stream(st):
                        st.wait_event(ready)
                        fn()
                for st in side:
                    main.wait_stream(st)

        def body_post():
            """after the exchange: every rank applies the identical NaN-check -> norm -> clip -> Adam (replicas stay
            bit-identical)"""
            self._opt_step("Vl", hp.lr_Vl)
            if not informarl:
                self._opt_step("Vh", hp.lr_Vh)
            self._opt_step("policy", hp.lr_actor)
            if lagr:
                lagr_step(apply_now=False)

        def exchange_lagr():
            # the `.mean()` of the multiplier step runs over the GLOBAL minibatch: all-reduce the n * nh sums, then every rank
            # applies the identical step (informarl_lagr.py:300-306)
            self.allreduce(self.lagr_sums)
            OA.lagr_apply(self.lagr, self.lagr_sums, self.world * Eb * T, hp.lr_lagr)

        def exchange():
            # data-parallel exchange: ONE all-reduce(sum) of [g_policy | g_Vl | g_Vh | loss sums] per minibatch
            # once the three backward passes have joined
            self.allreduce(self.flat_grads[:self.n_reduced])

        def step_body():
            body_pre()
            if reduce:
                exchange()
                body_post()
                if lagr:
                    exchange_lagr()

        # The minibatch step is ~400 launches of 10-100 us kernels: issuing them from Python costs about as much host time
        # as they take on the device.  With use_graphs the step is captured once into a HIP graph (all its operands live in
        # persistent buffers; only mb_idx changes) and replayed for every further minibatch and iteration.  Not with a
        # gradient hook (a Python callback).  With a gradient exchange the step is TWO graphs — everything before the
        # collective, and the optimiser steps after it — with the collective issued eagerly between the replays, so that no
        # communication library call is ever recorded into a graph.
        slot = None
        if self.use_graphs and is_cuda and self.grad_hook is None:
            key = (self.algo, B, Eb, ro.agent.data_ptr(), det.agent.data_ptr() if det is not None else 0,
                   tg["adv"].data_ptr(), tg["Ql"].data_ptr(), mb_idx.data_ptr(), self._update_generation())
            slot = self._upd_graph
            if slot.get("key") != key:
                slot.clear()
                slot["key"] = key
        for mb in range(n_mb):
            self._mb = mb
            mb_idx.copy_(idx_all[mb * Eb:(mb + 1) * Eb])
            mb_idx32.copy_(mb_idx)
            if slot is not None and slot.get("graph") is not None:
                slot["graph"].replay()
                if reduce:
                    exchange()
                    slot["graph_post"].replay()
                    if lagr:
                        exchange_lagr()
                K.FLOPS[0] += slot["flops"]
                continue
            f0 = K.FLOPS[0]
            step_body()
            if slot is not None and not slot.get("failed", False):
                slot["flops"] = K.FLOPS[0] - f0
                if slot.get("key")[-1] != self._update_generation():      # the eager pass grew a scratch buffer: try again later
                    slot["key"] = slot["key"][:-1] + (self._update_generation(),)
                    continue
                graph = torch.cuda.CUDAGraph()
                try:
                    f1 = K.FLOPS[0]
                    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                        body_pre()
                    if reduce:
                        gpost = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(gpost, capture_error_mode="thread_local"):
                            body_post()
                        slot["graph_post"] = gpost
                    K.FLOPS[0] = f1
                    slot["graph"] = graph
                except Exception as ex:                             # keep training: eager launches are always correct
                    slot["failed"] = True
                    torch.cuda.synchronize()
                    print(f"[dgppo_amd] HIP-graph capture of the minibatch step failed ({type(ex).__name__}: {ex}); "
                          f"continuing with eager launches", flush=True)
        th.append(_time.perf_counter())
        self._last = dict(Ql_mb=Ql_mb, G=G, R=R, nh=nh, B=B, Qh_mb=Qh_mb if lagr else None)
        out = self.info(ro)
        th.append(_time.perf_counter())
        # host time spent ISSUING finalize + targets, then the minibatch loop, then waiting for the device in info()
        self.host_ms = {"issue_targets": 1e3 * (th[1] - th[0]), "issue_minibatches": 1e3 * (th[2] - th[1]),
                        "wait_device": 1e3 * (th[3] - th[2])}
        return out

    def info(self, ro: RolloutData) -> dict:
        """scalars of the LAST minibatch (dgppo.py:292) with the reference's key names; one host sync.  Under data
        parallelism every value is GLOBAL: the loss / metric sums (stats rows 0..2) travelled with the gradients, the safe
        count and the target / log-pi extrema are reduced here over the host control plane (two tiny gloo collectives)."""
        from . import dist as D
        L = self._last
        s = self.stats.cpu().numpy().copy()
        n_bad = int(self.reset_failed.item())
        if n_bad:
            self.reset_failed.zero_()
            raise RuntimeError(f"env reset: {n_bad} environment(s) of the last rollouts got no valid scene within the kernels' "
                               f"rejection-loop bounds (too many agents / obstacles for the area?) — the batch is invalid")
        s[:3] /= self.world                                   # rows 0..2 were summed over the ranks with the gradients
        G, R, nh, B = L["G"], L["R"], L["nh"], L["B"]
        o = {k: self.opt[k].state[:8].cpu().numpy() for k in self.opt}
        pol_loss = s[2, 0] / R - self.hp.coef_ent * s[2, 1] / R
        lagr = self.algo == "informarl_lagr"
        ext = [float(L["Ql_mb"].max()), -float(L["Ql_mb"].min()), -float(ro.log_pis.min())]
        if lagr:
            ext += [float(L["Qh_mb"].max()), -float(L["Qh_mb"].min())]
        ext = D.host_allreduce(np.asarray(ext, np.float64), "max", self.world)
        safe = float(D.host_allreduce(np.asarray([s[3, 0]], np.float64), "sum", self.world)[0])
        out = {
            "Vl/loss": float(s[0, 0] / G), "Vl/grad_norm": float(o["Vl"][4]), "Vl/has_nan": float(o["Vl"][5]),
            "Vl/max_target": float(ext[0]), "Vl/min_target": float(-ext[1]),
            "policy/loss": float(pol_loss), "policy/grad_norm": float(o["policy"][4]), "policy/has_nan": float(o["policy"][5]),
            "policy/log_pi_min": float(-ext[2]), "policy/clip_frac": float(s[2, 2] / R),
            "policy/entropy": float(s[2, 1] / R), "policy/total_variation_dist": float(0.5 * s[2, 3] / R),
        }
        if lagr:   # informarl_lagr.py:278-282,309
            out.update({"Vh/loss": float(s[1, 0] / (R * nh)), "Vh/grad_norm": float(o["Vh"][4]), "Vh/has_nan": float(o["Vh"][5]),
                        "Vh/max_target": float(ext[3]), "Vh/min_target": float(-ext[4]),
                        "policy/lagr_mean": float(self.lagr.mean())})
        if self.algo == "dgppo":       # InforMARL logs only the Vl and policy keys (informarl.py:357-457)
            out.update({"Vh/loss_Vh": float(s[1, 0] / (R * nh)), "Vh/grad_Vh_norm": float(o["Vh"][4]),
                        "Vh/grad_Vh_has_nan": float(o["Vh"][5])})
        if self.algo in ("dgppo", "hcbfcrpo"):   # hcbfcrpo.py:204
            out["eval/safe_data"] = safe / (self.world * B * self.T * self.cfg.n_agents)
        return out
