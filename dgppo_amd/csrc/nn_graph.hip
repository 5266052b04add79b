// Graph side of the GraphTransformer layer (dgppo/nn/gnn.py:78-117) in per-agent fixed-fan-in form.
//
// Only agents receive messages and every sender slot of an agent has a static node id (SURVEY F5/F7;
// dgppo/utils/graph.py:35-44, dgppo/env/lidar_env/lidar_spread.py:57-96), so jraph.segment_softmax / segment_sum over
// the flat edge list reduce to a masked softmax over S = n + goal_slots + obs_slots slots per agent.  Masked edges are
// re-routed pad->pad in the reference and therefore never reach an agent: here they are simply excluded.
//
// Algebraic form (same function, different summation order; see DESIGN.md "GNN layer"):
//   logit[i,s,h] = <q_h(x_i), k_h(x_s)>/sqrt(D) = qt[i,h,:] . x_s + const(i,h)   with qt = x_i Mcat + c   (dense, outside)
//   sum_s a[i,s,h] (v_h(x_s) + e_h(edge_is)) = (sum_s a[i,s,h] [x_s ; edge_is]) [Wv_h ; We_h] + bv_h   (dense, outside)
// so this file only does: graph features, logits against RAW sender features, masked softmax, aggregation of raw
// features, and the matching backward.  const(i,h) (the key bias) cancels in the softmax.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

struct Topo {
  int n, ng, gs, os, per, lidar, spread;  // per = k (LiDAR) or n_obs (MPE)
  int S, Ns;                              // slots per agent, nodes without pad
};

static Topo make_topo(const dgppo_env_cfg& c) {
  Topo t;
  t.n = c.n_agents; t.ng = c.n_goals; t.gs = cfg_goal_slots(c); t.os = cfg_obs_slots(c);
  t.lidar = cfg_is_lidar(c) ? 1 : 0; t.spread = cfg_is_spread(c) ? 1 : 0;
  t.per = t.os;
  t.S = t.n + t.gs + t.os;
  t.Ns = t.n + t.ng + cfg_obs_nodes(c);
  return t;
}

__device__ inline int sender_node(const Topo& t, int i, int s) {
  if (s < t.n) return s;
  if (s < t.n + t.gs) return t.spread ? t.n + (s - t.n) : t.n + i;
  const int m = s - t.n - t.gs;
  return t.lidar ? t.n + t.ng + i * t.per + m : t.n + t.ng + m;
}
// slot through which node nd sends to agent i (or -1)
__device__ inline int slot_of(const Topo& t, int nd, int i) {
  if (nd < t.n) return nd;
  if (nd < t.n + t.ng) {
    const int g = nd - t.n;
    return t.spread ? t.n + g : (g == i ? t.n : -1);
  }
  const int q = nd - t.n - t.ng;
  if (t.lidar) return (q / t.per == i) ? t.n + t.gs + (q - i * t.per) : -1;
  return t.n + t.gs + q;
}

// ---------------------------------------------------------------------------------------------------------------------
// graph features: compact record -> agent/other node feature matrices, per-slot edge features and masks.
// Same arithmetic (explicit _rn ops = no fma contraction) as env_step.hip phase 5, i.e. as
// lidar_env/base.py:227-271 + lidar_spread.py:57-96 (+ MPE twins), so masks agree bit for bit with the GraphsTuple.
// ---------------------------------------------------------------------------------------------------------------------
struct FeatArgs {
  dgppo_env_cfg cfg;
  Topo t;
  const float* agent; long agent_se, agent_st;   // agent + env*se + time*st  -> [n, sd]
  const float* goal;                             // goal + env*ng*sd
  const float* obst;                             // MPE: obst + env*n_obs*sd
  const float* hits; long hits_se, hits_st;      // LiDAR: hits + env*se + time*st -> [n, k, 2]
  const int32_t* env_ids;                        // [n_env] or NULL (identity)
  int n_env, n_time;                             // graphs g = e * n_time + t
  float* Xa;      // [G*n, Fp]
  float* Xo;      // [G*(Ns-n), Fp]
  float* efeat;   // [G*n, S, 4]
  float* emask;   // [G*n, S]
  int Fp;
  uint32_t rcp_fp, rcp_S;   // ceil(2^32 / d): index divisions by Fp and S as one v_mul_hi_u32 (exact for idx < 2^16)
};

__device__ inline float dist_rn(float dx, float dy) { return __fsqrt_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy))); }

template <int SD>
__global__ void graph_feats_kernel(FeatArgs a) {
  extern __shared__ float sm[];
  const dgppo_env_cfg& c = a.cfg;
  const Topo& t = a.t;
  const int g = blockIdx.x;
  const int e = g / a.n_time, tt = g - e * a.n_time;
  const int env = a.env_ids ? a.env_ids[e] : e;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int n = t.n, ng = t.ng, S = t.S, Fp = a.Fp;
  const int n_on = t.Ns - n - ng;
  float* s_ag = sm;                 // n*SD
  float* s_go = s_ag + n * SD;      // ng*SD
  float* s_ob = s_go + ng * SD;     // n_on * 2 (positions of hit / obstacle nodes) then MPE extra state
  float* s_fa = s_ob + n_on * SD;   // n*4
  float* s_fg = s_fa + n * 4;       // ng*4
  const float* ag = a.agent + (size_t)env * a.agent_se + (size_t)tt * a.agent_st;
  for (int i = tid; i < n * SD; i += nt) s_ag[i] = ag[i];
  for (int i = tid; i < ng * SD; i += nt) s_go[i] = a.goal[(size_t)env * ng * SD + i];
  if (n_on > 0) {
    if (t.lidar) {
      const float* hp = a.hits + (size_t)env * a.hits_se + (size_t)tt * a.hits_st;
      for (int i = tid; i < n_on * 2; i += nt) {
        const int q = i >> 1, d = i & 1;
        s_ob[q * SD + d] = hp[i];
      }
      for (int i = tid; i < n_on * (SD - 2); i += nt) {
        const int q = i / (SD - 2), d = i - q * (SD - 2);
        s_ob[q * SD + 2 + d] = 0.0f;
      }
    } else {
      for (int i = tid; i < n_on * SD; i += nt) s_ob[i] = a.obst[(size_t)env * n_on * SD + i];
    }
  }
  __syncthreads();
  for (int i = tid; i < n + ng; i += nt) {
    const float* s = (i < n) ? s_ag + i * SD : s_go + (i - n) * SD;
    float* f = (i < n) ? s_fa + i * 4 : s_fg + (i - n) * 4;
    if constexpr (SD == 5) {
      f[0] = s[0]; f[1] = s[1]; f[2] = __fmul_rn(s[4], s[2]); f[3] = __fmul_rn(s[4], s[3]);
    } else {
      f[0] = s[0]; f[1] = s[1]; f[2] = s[2]; f[3] = s[3];
    }
  }
  __syncthreads();
  // node feature rows: [state | obs, goal, agent indicator], zero padded to Fp
  constexpr int ND = SD + 3;
  for (int idx = tid; idx < t.Ns * Fp; idx += nt) {
    const int nd = (int)__umulhi((uint32_t)idx, a.rcp_fp), col = idx - nd * Fp;
    float v = 0.0f;
    if (nd < n) v = (col < SD) ? s_ag[nd * SD + col] : ((col == SD + 2) ? 1.0f : 0.0f);
    else if (nd < n + ng) v = (col < SD) ? s_go[(nd - n) * SD + col] : ((col == SD + 1) ? 1.0f : 0.0f);
    else v = (col < SD) ? s_ob[(nd - n - ng) * SD + col] : ((col == SD) ? 1.0f : 0.0f);
    if (col >= ND) v = 0.0f;
    if (nd < n) a.Xa[((size_t)g * n + nd) * Fp + col] = v;
    else a.Xo[((size_t)g * (t.Ns - n) + (nd - n)) * Fp + col] = v;
  }
  // per-slot edge feature + mask
  for (int idx = tid; idx < n * S; idx += nt) {
    const int i = (int)__umulhi((uint32_t)idx, a.rcp_S), s = idx - i * S;
    float4 f;
    bool mask;
    const float* fi = s_fa + i * 4;
    const float px = s_ag[i * SD], py = s_ag[i * SD + 1];
    if (s < n) {
      const float* fj = s_fa + s * 4;
      f = make_float4(__fsub_rn(fi[0], fj[0]), __fsub_rn(fi[1], fj[1]), __fsub_rn(fi[2], fj[2]), __fsub_rn(fi[3], fj[3]));
      float d = __fadd_rn(dist_rn(__fsub_rn(px, s_ag[s * SD]), __fsub_rn(py, s_ag[s * SD + 1])), (i == s) ? c.eye_offset : 0.0f);
      mask = d < c.comm_radius;
    } else if (s < n + t.gs) {
      const int gi = t.spread ? (s - n) : i;
      const float* fg = s_fg + gi * 4;
      f = make_float4(__fsub_rn(fi[0], fg[0]), __fsub_rn(fi[1], fg[1]), __fsub_rn(fi[2], fg[2]), __fsub_rn(fi[3], fg[3]));
      mask = true;
    } else {
      const int m = s - n - t.gs;
      if (t.lidar) {
        const float* hp = s_ob + (i * t.per + m) * SD;
        const float lx = __fsub_rn(px, hp[0]), ly = __fsub_rn(py, hp[1]);
        f = make_float4(lx, ly, 0.0f, 0.0f);
        mask = dist_rn(lx, ly) < c.lidar_mask_radius;
      } else {
        const float* xo = s_ob + m * SD;
        const float* xi = s_ag + i * SD;
        f = make_float4(__fsub_rn(xi[0], xo[0]), __fsub_rn(xi[1], xo[1]), __fsub_rn(xi[2], xo[2]), __fsub_rn(xi[3], xo[3]));
        mask = dist_rn(__fsub_rn(xi[0], xo[0]), __fsub_rn(xi[1], xo[1])) < c.obs_mask_radius;   // mpe_corridor.py:93: 100 x comm_radius
      }
    }
    reinterpret_cast<float4*>(a.efeat)[(size_t)g * n * S + idx] = f;
    a.emask[(size_t)g * n * S + idx] = mask ? 1.0f : 0.0f;
  }
}

extern "C" int32_t dgppo_graph_feats(const dgppo_env_cfg* cfg, const float* agent, int64_t agent_se, int64_t agent_st,
                                     const float* goal, const float* obst, const float* hits, int64_t hits_se,
                                     int64_t hits_st, const int32_t* env_ids, int32_t n_env, int32_t n_time, float* Xa,
                                     float* Xo, float* efeat, float* emask, int32_t Fp, void* stream) {
  int32_t rc = dgppo_validate_cfg(cfg);
  if (rc) return rc;
  DGPPO_REQUIRE(n_env >= 0 && n_time >= 0, "graph_feats: negative counts");
  if (n_env == 0 || n_time == 0) return 0;
  DGPPO_REQUIRE(agent && goal && Xa && efeat && emask, "graph_feats: NULL operand");
  DGPPO_REQUIRE(Fp >= cfg->node_dim && Fp <= 32, "graph_feats: Fp must be in [node_dim, 32]");
  FeatArgs a;
  a.cfg = *cfg; a.t = make_topo(*cfg);
  const int n_on = a.t.Ns - a.t.n - a.t.ng;
  DGPPO_REQUIRE(n_on == 0 || Xo, "graph_feats: Xo is NULL");
  DGPPO_REQUIRE(a.t.Ns == a.t.n || Xo, "graph_feats: Xo is NULL");
  if (n_on > 0) {
    if (a.t.lidar) DGPPO_REQUIRE(hits, "graph_feats: hits is NULL");
    else DGPPO_REQUIRE(obst, "graph_feats: obst is NULL");
  }
  DGPPO_REQUIRE(((uintptr_t)efeat & 15) == 0, "graph_feats: efeat must be 16-byte aligned");
  a.agent = agent; a.agent_se = agent_se; a.agent_st = agent_st; a.goal = goal; a.obst = obst;
  a.hits = hits; a.hits_se = hits_se; a.hits_st = hits_st; a.env_ids = env_ids; a.n_env = n_env; a.n_time = n_time;
  a.Xa = Xa; a.Xo = Xo; a.efeat = efeat; a.emask = emask; a.Fp = Fp;
  a.rcp_fp = (uint32_t)((0x100000000ull + (uint64_t)Fp - 1) / (uint64_t)Fp);
  a.rcp_S = (uint32_t)((0x100000000ull + (uint64_t)a.t.S - 1) / (uint64_t)a.t.S);
  DGPPO_REQUIRE(Fp >= 2 && a.t.S >= 2 && (long)a.t.Ns * Fp < 65536 && (long)a.t.n * a.t.S < 65536, "graph_feats: sizes out of range");
  const int SD = cfg->state_dim;
  const size_t smem = sizeof(float) * ((size_t)a.t.n * SD + a.t.ng * SD + (size_t)n_on * SD + a.t.n * 4 + a.t.ng * 4);
  const long G = (long)n_env * n_time;
  if (SD == 5) hipLaunchKernelGGL(graph_feats_kernel<5>, dim3(G), dim3(128), smem, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(graph_feats_kernel<4>, dim3(G), dim3(128), smem, (hipStream_t)stream, a);
  DGPPO_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// attention forward: one workgroup per graph
// ---------------------------------------------------------------------------------------------------------------------
struct AttnArgs {
  Topo t;
  int F, H, Kp;          // feature width of this layer's inputs, heads, padded width of zcat
  const float* qt;       // [G*n, H*F]
  const float* Xa;       // [G*n, F]
  const float* Xo;       // [G*(Ns-n), F]
  const float* efeat;    // [G*n, S, 4]
  const float* emask;    // [G*n, S]
  float* zcat;           // [G*n, Kp]   = [x_i | per head: sum a x_s (F), sum a e (4) | 1 | 0-pad]
  float* attn;           // [G*n, S, H]
  // backward
  const float* dzcat;    // [G*n, Kp]
  float* dqt;            // [G*n, H*F]
  float* dXa;            // [G*n, F]   (written, not accumulated) or NULL
  float* dXo;            // [G*(Ns-n), F] or NULL
  int relu_xo;           // dXo *= (Xo > 0): Xo is the ReLU output of the previous layer's update (gnn.py:109-111, aggr = 0)
  int G;
  // block-diagonal kernels only (dgppo_attn_fwd_xo / _bwd_xo): the other nodes' rows are not read but recomputed,
  // Xo = relu(Xo_raw Wo + bo) with Xo_raw [G*(Ns-n), 8] the padded raw node features (gnn.py:109-111 with aggr = 0)
  const float* Xo_raw; const float* Wo; const float* bo; int ldwo;
  // backward with recomputed rows: instead of writing dXo, every graph writes raw^T dXo [8 x 32] and colsum dXo [32] (the gradient
  // of Wo / bo through the ReLU) to dwo_slab + g * ABD_DW_STRIDE; a reduce kernel adds the slabs up
  float* dwo_slab;
};

__global__ void __launch_bounds__(256) attn_fwd_valu_kernel(AttnArgs a) {
  extern __shared__ float sm[];
  const Topo& t = a.t;
  const int g = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const int n = t.n, S = t.S, Ns = t.Ns, F = a.F, H = a.H, Fl = F + 1;
  float* s_x = sm;                      // Ns * Fl
  float* s_q = s_x + Ns * Fl;           // n * H * Fl
  float* s_e = s_q + n * H * Fl;        // n * S * 4
  float* s_m = s_e + n * S * 4;         // n * S
  float* s_a = s_m + n * S;             // n * S * H
  for (int idx = tid; idx < Ns * F; idx += nt) {
    const int nd = idx / F, f = idx - nd * F;
    s_x[nd * Fl + f] = (nd < n) ? a.Xa[((size_t)g * n + nd) * F + f] : a.Xo[((size_t)g * (Ns - n) + (nd - n)) * F + f];
  }
  for (int idx = tid; idx < n * H * F; idx += nt) {
    const int ih = idx / F, f = idx - ih * F;
    s_q[ih * Fl + f] = a.qt[(size_t)g * n * H * F + idx];
  }
  for (int idx = tid; idx < n * S * 4; idx += nt) s_e[idx] = a.efeat[(size_t)g * n * S * 4 + idx];
  for (int idx = tid; idx < n * S; idx += nt) s_m[idx] = a.emask[(size_t)g * n * S + idx];
  __syncthreads();
  // logits
  for (int idx = tid; idx < n * S * H; idx += nt) {
    const int h = idx % H, is = idx / H;
    const int i = is / S, s = is - i * S;
    const float* x = s_x + sender_node(t, i, s) * Fl;
    const float* q = s_q + (i * H + h) * Fl;
    float acc = 0.0f;
    for (int f = 0; f < F; ++f) acc = fmaf(q[f], x[f], acc);
    s_a[idx] = acc;
  }
  __syncthreads();
  // masked softmax over the slots of (i, h)   [jraph.segment_softmax: subtract max, exp, normalise]
  for (int ih = tid; ih < n * H; ih += nt) {
    const int i = ih / H, h = ih - i * H;
    float mx = -INFINITY;
    for (int s = 0; s < S; ++s)
      if (s_m[i * S + s] != 0.0f) mx = fmaxf(mx, s_a[(i * S + s) * H + h]);
    float den = 0.0f;
    for (int s = 0; s < S; ++s) {
      float ev = 0.0f;
      if (s_m[i * S + s] != 0.0f) ev = expf(s_a[(i * S + s) * H + h] - mx);
      s_a[(i * S + s) * H + h] = ev;
      den += ev;
    }
    const float inv = (den > 0.0f) ? 1.0f / den : 0.0f;
    for (int s = 0; s < S; ++s) s_a[(i * S + s) * H + h] *= inv;
  }
  __syncthreads();
  if (a.attn != nullptr)
    for (int idx = tid; idx < n * S * H; idx += nt) a.attn[(size_t)g * n * S * H + idx] = s_a[idx];
  // aggregation of raw sender / edge features
  const int W = F + 4;
  for (int idx = tid; idx < n * H * W; idx += nt) {
    const int w = idx % W, ih = idx / W;
    const int i = ih / H, h = ih - i * H;
    float acc = 0.0f;
    if (w < F) {
      for (int s = 0; s < S; ++s) {
        const float av = s_a[(i * S + s) * H + h];
        if (av != 0.0f) acc = fmaf(av, s_x[sender_node(t, i, s) * Fl + w], acc);
      }
    } else {
      // masked slots carry a == 0; their (unclamped, up to 5e5) edge features must not produce 0*inf/NaN: skip them
      for (int s = 0; s < S; ++s) {
        const float av = s_a[(i * S + s) * H + h];
        if (av != 0.0f) acc = fmaf(av, s_e[(i * S + s) * 4 + (w - F)], acc);
      }
    }
    a.zcat[((size_t)g * n + i) * a.Kp + F + h * W + w] = acc;
  }
  for (int idx = tid; idx < n * F; idx += nt) {
    const int i = idx / F, f = idx - i * F;
    a.zcat[((size_t)g * n + i) * a.Kp + f] = s_x[i * Fl + f];
  }
  const int kc = F + H * W;  // index of the constant-one column (carries mean_h b_v)
  for (int idx = tid; idx < n * (a.Kp - kc); idx += nt) {
    const int i = idx / (a.Kp - kc), c = kc + idx - i * (a.Kp - kc);
    a.zcat[((size_t)g * n + i) * a.Kp + c] = (c == kc) ? 1.0f : 0.0f;
  }
}

__global__ void __launch_bounds__(256) attn_bwd_valu_kernel(AttnArgs a) {
  extern __shared__ float sm[];
  const Topo& t = a.t;
  const int g = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const int n = t.n, S = t.S, Ns = t.Ns, F = a.F, H = a.H, Fl = F + 1, W = F + 4;
  float* s_x = sm;                      // Ns * Fl
  float* s_q = s_x + Ns * Fl;           // n * H * Fl
  float* s_e = s_q + n * H * Fl;        // n * S * 4
  float* s_a = s_e + n * S * 4;         // n * S * H   attention weights
  float* s_d = s_a + n * S * H;         // n * S * H   dA then dlogit
  float* s_z = s_d + n * S * H;         // n * H * (W+1)  dz (aggregated part)
  const int Wl = W + 1;
  for (int idx = tid; idx < Ns * F; idx += nt) {
    const int nd = idx / F, f = idx - nd * F;
    s_x[nd * Fl + f] = (nd < n) ? a.Xa[((size_t)g * n + nd) * F + f] : a.Xo[((size_t)g * (Ns - n) + (nd - n)) * F + f];
  }
  for (int idx = tid; idx < n * H * F; idx += nt) {
    const int ih = idx / F, f = idx - ih * F;
    s_q[ih * Fl + f] = a.qt[(size_t)g * n * H * F + idx];
  }
  for (int idx = tid; idx < n * S * 4; idx += nt) s_e[idx] = a.efeat[(size_t)g * n * S * 4 + idx];
  for (int idx = tid; idx < n * S * H; idx += nt) s_a[idx] = a.attn[(size_t)g * n * S * H + idx];
  for (int idx = tid; idx < n * H * W; idx += nt) {
    const int w = idx % W, ih = idx / W;
    const int i = ih / H, h = ih - i * H;
    s_z[ih * Wl + w] = a.dzcat[((size_t)g * n + i) * a.Kp + F + h * W + w];
  }
  __syncthreads();
  // dA[i,s,h] = dzx[i,h,:] . x_s + dze[i,h,:] . e_is
  for (int idx = tid; idx < n * S * H; idx += nt) {
    const int h = idx % H, is = idx / H;
    const int i = is / S, s = is - i * S;
    float acc = 0.0f;
    if (s_a[idx] != 0.0f) {
      const float* x = s_x + sender_node(t, i, s) * Fl;
      const float* dz = s_z + (i * H + h) * Wl;
      for (int f = 0; f < F; ++f) acc = fmaf(dz[f], x[f], acc);
      for (int c = 0; c < 4; ++c) acc = fmaf(dz[F + c], s_e[is * 4 + c], acc);
    }
    s_d[idx] = acc;
  }
  __syncthreads();
  // softmax backward: dlogit = a * (dA - sum_s a dA)
  for (int ih = tid; ih < n * H; ih += nt) {
    const int i = ih / H, h = ih - i * H;
    float dot = 0.0f;
    for (int s = 0; s < S; ++s) dot = fmaf(s_a[(i * S + s) * H + h], s_d[(i * S + s) * H + h], dot);
    for (int s = 0; s < S; ++s) {
      const int k = (i * S + s) * H + h;
      s_d[k] = s_a[k] * (s_d[k] - dot);
    }
  }
  __syncthreads();
  // dqt[i,h,f] = sum_s dlogit[i,s,h] x_s[f]
  for (int idx = tid; idx < n * H * F; idx += nt) {
    const int f = idx % F, ih = idx / F;
    const int i = ih / H, h = ih - i * H;
    float acc = 0.0f;
    for (int s = 0; s < S; ++s) {
      const float dv = s_d[(i * S + s) * H + h];
      if (dv != 0.0f) acc = fmaf(dv, s_x[sender_node(t, i, s) * Fl + f], acc);
    }
    a.dqt[(size_t)g * n * H * F + idx] = acc;
  }
  // d x_nd[f] = sum over receivers i of sum_h (a * dzx + dlogit * qt)  (+ the direct x_i part for agents)
  if (a.dXa != nullptr) {
    for (int idx = tid; idx < Ns * F; idx += nt) {
      const int nd = idx / F, f = idx - nd * F;
      if (nd >= n && a.dXo == nullptr) continue;
      float acc = 0.0f;
      for (int i = 0; i < n; ++i) {
        const int s = slot_of(t, nd, i);
        if (s < 0) continue;
        for (int h = 0; h < H; ++h) {
          const int k = (i * S + s) * H + h;
          acc = fmaf(s_a[k], s_z[(i * H + h) * Wl + f], acc);
          acc = fmaf(s_d[k], s_q[(i * H + h) * Fl + f], acc);
        }
      }
      if (nd < n) a.dXa[((size_t)g * n + nd) * F + f] = acc + a.dzcat[((size_t)g * n + nd) * a.Kp + f];
      else a.dXo[((size_t)g * (Ns - n) + (nd - n)) * F + f] = acc;
    }
  }
}


// =====================================================================================================================
// Matrix-core attention.  Per graph (one workgroup, 4 waves) everything is a small dense product through LDS:
//   L [nH x Ns]  = Qt [nH x F] * Xs^T            logits of every (agent, head) against every node      (MFMA)
//   a            = masked softmax of L gathered at the agent's S slots                                   (half-wave shuffles)
//   P [nH x Ns]  = a scattered back to node columns (0 elsewhere)
//   Zx [nH x F]  = P * Xs                          aggregated sender features                            (MFMA)
//   ze [nH x 4]  = sum_s a * edge feature                                                                 (VALU, tiny)
// backward:  dA = dZx * Xs^T (MFMA) (+ dze . e), softmax backward, dQt = dL * Xs (MFMA),
//            dXs = P^T * dZx + dL^T * Qt (MFMA), agents also get the direct x_i part of dzcat.
// v_mfma_f32_16x16x4_f32: A[i = lane&15][k = lane>>4], B[k = lane>>4][j = lane&15], D[row = (lane>>4)*4 + r][col = lane&15].
// Padding rows/columns of every operand are zero-filled so that they cannot leak into valid outputs.
// =====================================================================================================================
using f32x4g = __attribute__((ext_vector_type(4))) float;
#ifdef DGPPO_STAMPS
__device__ unsigned long long g_astamps[32];
#define ASTAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_astamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int32_t dgppo_debug_stamps_attn(unsigned long long* out) {
  return (int32_t)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_astamps), sizeof(unsigned long long) * 32);
}
#else
#define ASTAMP(i)
#endif

struct AttnDims {
  int n, H, F, S, Ns, nH, RT, CT, Fl, Ll, W;   // RT = ceil(nH/16), CT = ceil(Ns/16), W = lanes per (agent, head) pair
};
__host__ __device__ inline AttnDims attn_dims(const Topo& t, int F, int H) {
  AttnDims d;
  d.n = t.n; d.H = H; d.F = F; d.S = t.S; d.Ns = t.Ns; d.nH = t.n * H;
  d.RT = (d.nH + 15) / 16; d.CT = (d.Ns + 15) / 16;
  d.Fl = F + 1; d.Ll = d.CT * 16 + 1; d.W = (t.S > 32) ? 64 : 32;
  return d;
}
static size_t attn_mfma_smem(const AttnDims& d, bool bwd) {
  size_t fl = (size_t)d.CT * 16 * d.Fl        // s_x   node features (rows >= Ns zero)
              + (size_t)d.RT * 16 * d.Fl      // s_q   qt (rows >= nH zero)
              + (size_t)d.RT * 16 * d.Ll      // s_L   logits -> P
              + 8 + (size_t)d.n * d.S * 5     // s_e (4, float4-aligned) + s_m (1)
              + 2 * (size_t)d.nH * ((d.S + 3) & ~3);   // compact per-pair rows (s_a, and s_c in the backward)
  if (bwd) fl += (size_t)d.RT * 16 * d.Ll     // s_D   dA -> dL (dense)
                 + (size_t)d.RT * 16 * (d.F + 5);  // s_z   dz (aggregated part), rows >= nH zero
  return fl * sizeof(float);
}

__device__ inline float red_max(float v, int W) {
  for (int o = W >> 1; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, W));
  return v;
}
__device__ inline float red_sum(float v, int W) {
  for (int o = W >> 1; o >= 1; o >>= 1) v += __shfl_xor(v, o, W);
  return v;
}

// C[rows x cols] (tile list dealt round-robin to the 4 waves) = A[rows x K] * B[K x cols]; element accessors are lambdas
template <typename FA, typename FB, typename FC>
__device__ inline void mfma_tiles(int RTn, int CTn, int K4, int wave, int lane, FA fa, FB fb, FC fc) {
  const int li = lane & 15, lq = lane >> 4;
  for (int tile = wave; tile < RTn * CTn; tile += 4) {
    const int rt = tile / CTn, ct = tile - rt * CTn;
    f32x4g acc = {0.f, 0.f, 0.f, 0.f};
    // fragments of 4 k-steps are fetched together (one LDS round trip), then 4 MFMAs issue back to back
    int k4 = 0;
    for (; k4 + 4 <= K4; k4 += 4) {
      float av[4], bv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { av[u] = fa(rt * 16 + li, (k4 + u) * 4 + lq); bv[u] = fb((k4 + u) * 4 + lq, ct * 16 + li); }
#pragma unroll
      for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
    }
    for (; k4 < K4; ++k4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa(rt * 16 + li, k4 * 4 + lq), fb(k4 * 4 + lq, ct * 16 + li), acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) fc(rt * 16 + lq * 4 + r, ct * 16 + li, acc[r]);
  }
}


// Issue up to MAXIT float4 loads per lane back to back (all in flight together), then hand them to `put`.
template <int MAXIT, typename FL, typename FP>
__device__ inline void stage4(int count4, int tid, FL load, FP put) {
  float4 v[MAXIT];
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    const int idx = tid + it * 256;
    v[it] = (idx < count4) ? load(idx) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    const int idx = tid + it * 256;
    if (idx < count4) put(idx, v[it]);
  }
  for (int idx = tid + MAXIT * 256; idx < count4; idx += 256) put(idx, load(idx));
}
__device__ inline void put4(float* d, float4 v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w; }

#define ATT_SMAX 64   // max slots per agent handled by the per-pair register row

__global__ void __launch_bounds__(256) attn_fwd_kernel(AttnArgs a) {
  extern __shared__ float sm[];
  const Topo& t = a.t;
  const AttnDims d = attn_dims(t, a.F, a.H);
  const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = d.n, S = d.S, Ns = d.Ns, F = d.F, H = d.H, Fl = d.Fl, Ll = d.Ll, nH = d.nH;
  const int Sp = (S + 3) & ~3;
  float* s_x = sm;
  float* s_q = s_x + d.CT * 16 * Fl;
  float* s_L = s_q + d.RT * 16 * Fl;
  float* s_e = sm + (((d.CT * 16 * Fl + d.RT * 16 * Fl + d.RT * 16 * Ll) + 3) & ~3);   // float4-aligned
  float* s_m = s_e + n * S * 4;
  float* s_a = s_m + n * S;               // compact [nH][Sp]: logits at the slots -> attention weights
  ASTAMP(0);
  // ---- stage operands (zero padding): every global load of the lane is issued before the first LDS store ----
  {
    const int F4 = F >> 2;
    const float4* xa = reinterpret_cast<const float4*>(a.Xa + (size_t)g * n * F);
    const float4* xo = reinterpret_cast<const float4*>(a.Xo + (size_t)g * (Ns - n) * F);
    const float4* q4 = reinterpret_cast<const float4*>(a.qt + (size_t)g * nH * F);
    const float4* e4 = reinterpret_cast<const float4*>(a.efeat + (size_t)g * n * S * 4);
    const float* mk = a.emask + (size_t)g * n * S;
    const int nA = n * F4, nX = Ns * F4, nXp = d.CT * 16 * F4, nQ = nH * F4, nQp = d.RT * 16 * F4, nE = n * S;
    float4 vx[4], vq[2], ve[2];
    float vm[2];
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int it = 0; it < 4; ++it) { const int idx = tid + it * 256; vx[it] = (idx < nA) ? xa[idx] : ((idx < nX) ? xo[idx - nA] : z4); }
#pragma unroll
    for (int it = 0; it < 2; ++it) { const int idx = tid + it * 256; vq[it] = (idx < nQ) ? q4[idx] : z4; ve[it] = (idx < nE) ? e4[idx] : z4; vm[it] = (idx < nE) ? mk[idx] : 0.0f; }
#pragma unroll
    for (int it = 0; it < 4; ++it) { const int idx = tid + it * 256; if (idx < nXp) { const int nd = idx / F4, q = idx - nd * F4; put4(s_x + nd * Fl + 4 * q, vx[it]); } }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int idx = tid + it * 256;
      if (idx < nQp) { const int row = idx / F4, q = idx - row * F4; put4(s_q + row * Fl + 4 * q, vq[it]); }
      if (idx < nE) { const float4 ev = ve[it]; reinterpret_cast<float4*>(s_e)[idx] = make_float4(ev.x, ev.y, ev.z, ev.w); s_m[idx] = vm[it]; }
    }
    // sizes beyond the register budget (large n): plain loops
    for (int idx = tid + 1024; idx < nXp; idx += 256) { const int nd = idx / F4, q = idx - nd * F4; put4(s_x + nd * Fl + 4 * q, (idx < nA) ? xa[idx] : ((idx < nX) ? xo[idx - nA] : z4)); }
    for (int idx = tid + 512; idx < nQp; idx += 256) { const int row = idx / F4, q = idx - row * F4; put4(s_q + row * Fl + 4 * q, (idx < nQ) ? q4[idx] : z4); }
    for (int idx = tid + 512; idx < nE; idx += 256) { reinterpret_cast<float4*>(s_e)[idx] = e4[idx]; s_m[idx] = mk[idx]; }
  }
  __syncthreads();
  ASTAMP(1);
  // ---- L = Qt * Xs^T ----
  mfma_tiles(d.RT, d.CT, F / 4, wave, lane,
             [&](int row, int k) { return s_q[row * Fl + k]; },
             [&](int k, int col) { return s_x[col * Fl + k]; },
             [&](int row, int col, float v) { s_L[row * Ll + col] = v; });
  __syncthreads();
  ASTAMP(2);
  // ---- gather the logits at the agent's slots into the compact rows (masked slots -> -inf) ----
  for (int idx = tid; idx < nH * Sp; idx += 256) {
    const int pair = idx / Sp, s = idx - pair * Sp, i = pair / H;
    float l = -INFINITY;
    if (s < S && s_m[i * S + s] != 0.0f) l = s_L[pair * Ll + sender_node(t, i, s)];
    s_a[idx] = l;
  }
  __syncthreads();
  ASTAMP(3);
  // ---- 8 lanes per (agent, head): softmax over the compact row (<= 8 slots per lane in registers, 3-step xor
  //      reductions) + the edge-feature aggregation; afterwards every lane helps clearing s_L for the scatter of P ----
  const int Wd = F + 4;
  float* zc = a.zcat + (size_t)g * n * a.Kp;
  for (int p0 = 0; p0 < nH; p0 += 32) {
    const int pair = p0 + (tid >> 3), sub = tid & 7;
    const bool live = pair < nH;
    const int i = live ? pair / H : 0, h = live ? pair - (pair / H) * H : 0;
    float l[ATT_SMAX / 8];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < ATT_SMAX / 8; ++j) {
      const int sl = sub + 8 * j;
      l[j] = (live && sl < S) ? s_a[pair * Sp + sl] : -INFINITY;
      mx = fmaxf(mx, l[j]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 1, 8)); mx = fmaxf(mx, __shfl_xor(mx, 2, 8)); mx = fmaxf(mx, __shfl_xor(mx, 4, 8));
    float den = 0.0f;
#pragma unroll
    for (int j = 0; j < ATT_SMAX / 8; ++j) {
      if (sub + 8 * j < S) { const float ev = (l[j] == -INFINITY) ? 0.0f : expf(l[j] - mx); l[j] = ev; den += ev; }
    }
    den += __shfl_xor(den, 1, 8); den += __shfl_xor(den, 2, 8); den += __shfl_xor(den, 4, 8);
    const float inv = (den > 0.0f) ? 1.0f / den : 0.0f;
    float z0 = 0.f, z1 = 0.f, z2 = 0.f, z3 = 0.f;
#pragma unroll
    for (int j = 0; j < ATT_SMAX / 8; ++j) {
      const int sl = sub + 8 * j;
      if (live && sl < S) {
        const float av = l[j] * inv;
        s_a[pair * Sp + sl] = av;
        if (av != 0.0f) {   // masked slots may carry 5e5 / NaN edge features: skip, never multiply
          const float4 e = reinterpret_cast<const float4*>(s_e)[i * S + sl];
          z0 = fmaf(av, e.x, z0); z1 = fmaf(av, e.y, z1); z2 = fmaf(av, e.z, z2); z3 = fmaf(av, e.w, z3);
        }
      }
    }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) { z0 += __shfl_xor(z0, o, 8); z1 += __shfl_xor(z1, o, 8); z2 += __shfl_xor(z2, o, 8); z3 += __shfl_xor(z3, o, 8); }
    if (live && sub == 0) *reinterpret_cast<float4*>(zc + i * a.Kp + F + h * Wd + F) = make_float4(z0, z1, z2, z3);
  }
  for (int idx = tid; idx < d.RT * 16 * Ll; idx += 256) s_L[idx] = 0.0f;
  __syncthreads();
  ASTAMP(4);
  // ---- scatter P, write the attention weights ----
  for (int idx = tid; idx < n * S * H; idx += 256) {
    const int h = idx % H, is = idx / H, i = is / S, s = is - i * S;
    const float av = s_a[(i * H + h) * Sp + s];
    if (a.attn != nullptr) a.attn[(size_t)g * n * S * H + idx] = av;
    if (av != 0.0f) s_L[(i * H + h) * Ll + sender_node(t, i, s)] = av;
  }
  __syncthreads();
  ASTAMP(5);
  // ---- Zx = P * Xs, written straight into zcat; the direct x_i part and the constant column ----
  mfma_tiles(d.RT, (F + 15) / 16, d.CT * 4, wave, lane,
             [&](int row, int k) { return s_L[row * Ll + k]; },
             [&](int k, int col) { return (col < F) ? s_x[k * Fl + col] : 0.0f; },
             [&](int row, int col, float v) {
               if (row < nH && col < F) { const int i = row / H, h = row - i * H; zc[i * a.Kp + F + h * Wd + col] = v; }
             });
  for (int idx = tid; idx < n * F; idx += 256) {
    const int i = idx / F, f = idx - i * F;
    zc[i * a.Kp + f] = s_x[i * Fl + f];
  }
  const int kc = F + H * Wd;
  for (int idx = tid; idx < n * (a.Kp - kc); idx += 256) {
    const int i = idx / (a.Kp - kc), c = kc + idx - i * (a.Kp - kc);
    zc[i * a.Kp + c] = (c == kc) ? 1.0f : 0.0f;
  }
  ASTAMP(6);
}

// reductions over the 8 lanes that own an (agent, head) pair: DPP moves inside the VALU (quad swaps, then the mirror of
// the 8-lane half row); every lane of the group ends with the result.  __shfl_xor(width 8) would be 3 LDS-crossbar round
// trips per reduction.
template <int CTRL> __device__ inline float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ inline float grp8_sum(float v) {
  v += dpp_f<0xB1>(v);     // quad_perm [1,0,3,2]
  v += dpp_f<0x4E>(v);     // quad_perm [2,3,0,1]
  v += dpp_f<0x141>(v);    // row_half_mirror
  return v;
}
__device__ inline float grp8_max(float v) {
  v = fmaxf(v, dpp_f<0xB1>(v));
  v = fmaxf(v, dpp_f<0x4E>(v));
  v = fmaxf(v, dpp_f<0x141>(v));
  return v;
}

// ---- narrow layers (F = 8: the first GNN layer, node features padded 7 -> 8): slot-sparse VALU kernels ---------------
// With 8 features a dense [n*H x nodes] logit tile on the matrix cores computes 3.3x more products than the S slots of
// an agent need (24 of 80 nodes for LidarSpread n = 8) and needs three dependent memory round trips per graph (operands
// -> logits through LDS -> P -> aggregation).  Here one wave owns a graph and a group of 8 lanes owns an AGENT: every
// lane loads the sender rows / edge features / masks of its SJ slots STRAIGHT from global memory, once for all heads
// (sender ids are static, so every address is known at entry: ONE memory round trip per graph), forms its H * SJ logits
// with 8 FMAs each, and the per-(agent, head) reductions are DPP row operations.  No LDS, no matrix cores, ~1/3 of the
// instructions and of the L1 requests; sums run over slots in a different order than the dense form (tests: 1e-5).
// NPA = ceil(n / 8) passes over groups of 8 agents.
template <int H, int NPA, int SJ>
__global__ void __launch_bounds__(256) attn_fwd_slot8_kernel(AttnArgs a) {
  constexpr int F = 8, Wd = F + 4;
  const Topo& t = a.t;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane & 7, grp = lane >> 3;
  const int g = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave);
  if (g >= a.G) return;
  const int n = t.n, S = t.S, Ns = t.Ns, Kp = a.Kp, kc = F + H * Wd;
  const float* Xa = a.Xa + (size_t)g * n * F;
  const float* Xo = a.Xo + (size_t)g * (Ns - n) * F - (size_t)n * F;   // indexed by node id (>= n)
  const float* qt = a.qt + (size_t)g * n * H * F;
  const float* ef = a.efeat + (size_t)g * n * S * 4;
  const float* mk = a.emask + (size_t)g * n * S;
  float* zc = a.zcat + (size_t)g * n * Kp;
  float* at = a.attn + (size_t)g * n * S * H;
  // the parts of zcat that are plain copies: x_i, the constant column, zero padding
  for (int idx = lane; idx < n * 2; idx += 64)
    *reinterpret_cast<float4*>(zc + (idx >> 1) * Kp + 4 * (idx & 1)) = reinterpret_cast<const float4*>(Xa)[idx];
  for (int i = lane; i < n; i += 64)
    for (int c = kc; c < Kp; ++c) zc[i * Kp + c] = (c == kc) ? 1.0f : 0.0f;
#pragma unroll
  for (int p = 0; p < NPA; ++p) {
    const int ia = grp + 8 * p;
    const bool live = ia < n;
    const int i = live ? ia : n - 1;
    float4 q[H][2], x[SJ][2], e[SJ];
    float m[SJ];
#pragma unroll
    for (int h = 0; h < H; ++h) {
      const float4* qp = reinterpret_cast<const float4*>(qt + (i * H + h) * F);
      q[h][0] = qp[0]; q[h][1] = qp[1];
    }
#pragma unroll
    for (int j = 0; j < SJ; ++j) {
      int sl = sub + 8 * j;
      sl = sl < S ? sl : S - 1;
      const int nd = sender_node(t, i, sl);
      const float4* xp = reinterpret_cast<const float4*>((nd < n ? Xa : Xo) + (size_t)nd * F);
      x[j][0] = xp[0]; x[j][1] = xp[1];
      e[j] = reinterpret_cast<const float4*>(ef)[i * S + sl];
      m[j] = mk[i * S + sl];
    }
    float av[SJ][H];
#pragma unroll
    for (int h = 0; h < H; ++h) {
      float mx = -INFINITY;
#pragma unroll
      for (int j = 0; j < SJ; ++j) {
        const bool ok = live && (sub + 8 * j) < S && m[j] != 0.0f;
        float acc = q[h][0].x * x[j][0].x;
        acc = fmaf(q[h][0].y, x[j][0].y, acc); acc = fmaf(q[h][0].z, x[j][0].z, acc); acc = fmaf(q[h][0].w, x[j][0].w, acc);
        acc = fmaf(q[h][1].x, x[j][1].x, acc); acc = fmaf(q[h][1].y, x[j][1].y, acc);
        acc = fmaf(q[h][1].z, x[j][1].z, acc); acc = fmaf(q[h][1].w, x[j][1].w, acc);
        av[j][h] = ok ? acc : -INFINITY;
        mx = fmaxf(mx, av[j][h]);
      }
      mx = grp8_max(mx);
      float den = 0.0f;
#pragma unroll
      for (int j = 0; j < SJ; ++j) {
        const float ev = (av[j][h] == -INFINITY) ? 0.0f : __expf(av[j][h] - mx);   /* v_exp_f32: rel. error ~1e-7 on weights <= 1 */
        av[j][h] = ev;
        den += ev;
      }
      den = grp8_sum(den);
      const float inv = (den > 0.0f) ? 1.0f / den : 0.0f;
#pragma unroll
      for (int j = 0; j < SJ; ++j) av[j][h] *= inv;
    }
#pragma unroll
    for (int j = 0; j < SJ; ++j) {
      const int sl = sub + 8 * j;
      if (live && sl < S) {
#pragma unroll
        for (int h = 0; h < H; ++h) if (a.attn != nullptr) at[(i * S + sl) * H + h] = av[j][h];
      }
    }
#pragma unroll
    for (int h = 0; h < H; ++h) {
      float z[Wd];
#pragma unroll
      for (int c = 0; c < Wd; ++c) z[c] = 0.0f;
#pragma unroll
      for (int j = 0; j < SJ; ++j) {
        const float w = av[j][h];
        if (w != 0.0f) {   // masked slots (and dead lanes) have w == 0; they may carry 5e5 / NaN features: skip, never multiply
          z[0] = fmaf(w, x[j][0].x, z[0]); z[1] = fmaf(w, x[j][0].y, z[1]); z[2] = fmaf(w, x[j][0].z, z[2]); z[3] = fmaf(w, x[j][0].w, z[3]);
          z[4] = fmaf(w, x[j][1].x, z[4]); z[5] = fmaf(w, x[j][1].y, z[5]); z[6] = fmaf(w, x[j][1].z, z[6]); z[7] = fmaf(w, x[j][1].w, z[7]);
          z[8] = fmaf(w, e[j].x, z[8]); z[9] = fmaf(w, e[j].y, z[9]); z[10] = fmaf(w, e[j].z, z[10]); z[11] = fmaf(w, e[j].w, z[11]);
        }
      }
#pragma unroll
      for (int c = 0; c < Wd; ++c) z[c] = grp8_sum(z[c]);
      if (live && sub == h) {             // lane h of the group stores head h: the stores of the H heads go out together
        float4* o = reinterpret_cast<float4*>(zc + i * Kp + F + h * Wd);   // (F + h*Wd) % 4 == 0 and Kp % 4 == 0
        o[0] = make_float4(z[0], z[1], z[2], z[3]);
        o[1] = make_float4(z[4], z[5], z[6], z[7]);
        o[2] = make_float4(z[8], z[9], z[10], z[11]);
      }
    }
  }
}

// backward of the above for the first layer, whose inputs are raw features (no dXa / dXo): dqt only.
//   dA = dzx . x_s + dze . e  at the slots with a != 0;  dl = a (dA - sum_s a dA);  dqt[i,h,:] = sum_s dl x_s
template <int H, int NPA, int SJ>
__global__ void __launch_bounds__(256) attn_bwd_slot8_kernel(AttnArgs a) {
  constexpr int F = 8, Wd = F + 4;
  const Topo& t = a.t;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane & 7, grp = lane >> 3;
  const int g = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave);
  if (g >= a.G) return;
  const int n = t.n, S = t.S, Ns = t.Ns, Kp = a.Kp;
  const float* Xa = a.Xa + (size_t)g * n * F;
  const float* Xo = a.Xo + (size_t)g * (Ns - n) * F - (size_t)n * F;
  const float* ef = a.efeat + (size_t)g * n * S * 4;
  const float* at = a.attn + (size_t)g * n * S * H;
  const float* dzc = a.dzcat + (size_t)g * n * Kp;
  float* dq = a.dqt + (size_t)g * n * H * F;
#pragma unroll
  for (int p = 0; p < NPA; ++p) {
    const int ia = grp + 8 * p;
    const bool live = ia < n;
    const int i = live ? ia : n - 1;
    float4 dz[H][3], x[SJ][2], e[SJ];
    float av[SJ][H];
#pragma unroll
    for (int h = 0; h < H; ++h) {
      const float4* zp = reinterpret_cast<const float4*>(dzc + i * Kp + F + h * Wd);
      dz[h][0] = zp[0]; dz[h][1] = zp[1]; dz[h][2] = zp[2];
    }
#pragma unroll
    for (int j = 0; j < SJ; ++j) {
      int sl = sub + 8 * j;
      const bool in = live && sl < S;
      sl = sl < S ? sl : S - 1;
      const int nd = sender_node(t, i, sl);
      const float4* xp = reinterpret_cast<const float4*>((nd < n ? Xa : Xo) + (size_t)nd * F);
      x[j][0] = xp[0]; x[j][1] = xp[1];
      e[j] = reinterpret_cast<const float4*>(ef)[i * S + sl];
#pragma unroll
      for (int h = 0; h < H; ++h) {
        const float w = at[(i * S + sl) * H + h];
        av[j][h] = in ? w : 0.0f;
      }
    }
#pragma unroll
    for (int h = 0; h < H; ++h) {
      float dA[SJ];
      float dot = 0.0f;
#pragma unroll
      for (int j = 0; j < SJ; ++j) {
        float acc = 0.0f;
        if (av[j][h] != 0.0f) {               // masked slots may carry 5e5 / NaN features: skip, never multiply
          acc = dz[h][0].x * x[j][0].x;
          acc = fmaf(dz[h][0].y, x[j][0].y, acc); acc = fmaf(dz[h][0].z, x[j][0].z, acc); acc = fmaf(dz[h][0].w, x[j][0].w, acc);
          acc = fmaf(dz[h][1].x, x[j][1].x, acc); acc = fmaf(dz[h][1].y, x[j][1].y, acc);
          acc = fmaf(dz[h][1].z, x[j][1].z, acc); acc = fmaf(dz[h][1].w, x[j][1].w, acc);
          acc = fmaf(dz[h][2].x, e[j].x, acc); acc = fmaf(dz[h][2].y, e[j].y, acc);
          acc = fmaf(dz[h][2].z, e[j].z, acc); acc = fmaf(dz[h][2].w, e[j].w, acc);
          dot = fmaf(av[j][h], acc, dot);
        }
        dA[j] = acc;
      }
      dot = grp8_sum(dot);
      float o[F];
#pragma unroll
      for (int c = 0; c < F; ++c) o[c] = 0.0f;
#pragma unroll
      for (int j = 0; j < SJ; ++j) {
        if (av[j][h] != 0.0f) {
          const float dl = av[j][h] * (dA[j] - dot);
          o[0] = fmaf(dl, x[j][0].x, o[0]); o[1] = fmaf(dl, x[j][0].y, o[1]); o[2] = fmaf(dl, x[j][0].z, o[2]); o[3] = fmaf(dl, x[j][0].w, o[3]);
          o[4] = fmaf(dl, x[j][1].x, o[4]); o[5] = fmaf(dl, x[j][1].y, o[5]); o[6] = fmaf(dl, x[j][1].z, o[6]); o[7] = fmaf(dl, x[j][1].w, o[7]);
        }
      }
#pragma unroll
      for (int c = 0; c < F; ++c) o[c] = grp8_sum(o[c]);
      if (live && sub == h) {
        float4* op = reinterpret_cast<float4*>(dq + (i * H + h) * F);
        op[0] = make_float4(o[0], o[1], o[2], o[3]);
        op[1] = make_float4(o[4], o[5], o[6], o[7]);
      }
    }
  }
}

template <int H, int NPA>
static bool launch_attn_slot8_sj(const AttnArgs& a, int SJ, int grid, hipStream_t s, bool bwd) {
#define DGPPO_SJ(J)                                                                                              \
  case J:                                                                                                        \
    if (bwd) hipLaunchKernelGGL((attn_bwd_slot8_kernel<H, NPA, J>), dim3(grid), dim3(256), 0, s, a);             \
    else hipLaunchKernelGGL((attn_fwd_slot8_kernel<H, NPA, J>), dim3(grid), dim3(256), 0, s, a);                 \
    return true;
  switch (SJ) {
    DGPPO_SJ(1) DGPPO_SJ(2) DGPPO_SJ(3) DGPPO_SJ(4) DGPPO_SJ(5) DGPPO_SJ(6) DGPPO_SJ(7) DGPPO_SJ(8)
    default: return false;
  }
#undef DGPPO_SJ
}
// H = 3 heads (the reference's GraphTransformer default, dgppo/nn/gnn.py:81), n <= 32 agents, S <= 64 slots
static bool launch_attn_slot8(const AttnArgs& a, int grid, hipStream_t s, bool bwd) {
  if (a.H != 3 || a.H > 8) return false;
  const int NPA = (a.t.n + 7) / 8, SJ = (a.t.S + 7) / 8;
  switch (NPA) {
    case 1: return launch_attn_slot8_sj<3, 1>(a, SJ, grid, s, bwd);
    case 2: return launch_attn_slot8_sj<3, 2>(a, SJ, grid, s, bwd);
    case 3: return launch_attn_slot8_sj<3, 3>(a, SJ, grid, s, bwd);
    case 4: return launch_attn_slot8_sj<3, 4>(a, SJ, grid, s, bwd);
    default: return false;
  }
}

// ---- one wave per graph --------------------------------------------------------------------------------------------
// For the graph sizes DGPPO uses (n*H <= 32 query rows, <= 96 nodes) a whole graph fits one wave: there is no
// workgroup barrier anywhere, the 4 waves of a workgroup run 4 independent graphs and other waves fill the stalls.
//   * MFMA fragments come straight from global memory in fragment layout.  The sum over features may visit k in any
//     order as long as A and B agree, so lane (li, lq) takes the F/4 CONTIGUOUS features lq*F/4 .. of row li: a few
//     16-byte loads per lane instead of strided 4-byte ones.
//   * only the logit tile lives in LDS: L = Qt Xs^T is written there, the 8 lanes that own an (agent, head) pair read
//     its logits, and overwrite the row with P in place (DS operations of one wave are ordered), then Zx = P Xs reads
//     P as the A operand while the B operand (node rows, already in L2) is loaded directly in fragment layout.
#ifndef DGPPO_ATTN_BWD_WPE
#define DGPPO_ATTN_BWD_WPE 3
#endif
#define ATW_RT 2      // row tiles (n*H <= 32)
#define ATW_BX 48     // registers for the node fragments of the logit GEMM: CT * F/4 <= 48
#define ATW_BZ 48     // registers for the node fragments of the aggregation GEMM: 4*CT * ceil(F/16) <= 48
// CT = ceil(Ns/16) node tiles, NP = ceil(n*H/8) softmax passes of 8 (agent, head) pairs, SJ = ceil(S/8) slots per softmax
// lane: compile-time, so the kernel is straight-line code and the per-slot mask / edge features can be prefetched into
// registers.
template <int F, int CT, int NP, int SJ>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 8))) attn_fwd_wave_kernel(AttnArgs a) {
  extern __shared__ float sm[];
  constexpr int FQ = F / 4, FT = (F + 15) / 16, KZ = CT * 4, Ll = CT * 16 + 1, RT = (NP + 1) / 2;   // RT query-row tiles
  static_assert(CT * FQ <= ATW_BX && KZ * FT <= ATW_BZ && RT <= ATW_RT, "fragments exceed the register budget");
  const Topo& t = a.t;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lq = lane >> 4;
  // wave-uniform by construction; readfirstlane tells the compiler, so the per-graph base pointers live in SGPRs and the
  // loads use scalar-base + 32-bit-offset addressing instead of 64-bit vector address arithmetic
  const int g = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave);
  if (g >= a.G) return;                                        // no barriers below: a wave may leave on its own
  const int n = t.n, S = t.S, Ns = t.Ns, H = a.H, nH = n * H, Kp = a.Kp;
  const int Wd = F + 4, kc = F + H * Wd;
  const int hmagic = (65536 + H - 1) / H;                      // row / H == (row * hmagic) >> 16 for row * H < 65536
  float* s_L = sm + wave * (RT * 16 * Ll);
  ASTAMP(0);
  const float* Xa = a.Xa + (size_t)g * n * F;
  const float* Xo = a.Xo + (size_t)g * (Ns - n) * F - (size_t)n * F;   // indexed by node id (>= n)
  const float* qt = a.qt + (size_t)g * nH * F;
  float* zc = a.zcat + (size_t)g * n * Kp;
  auto xrow = [&](int node) -> const float* {                  // rows past Ns are clamped: they only meet zeros of P
    node = node < Ns ? node : Ns - 1;
    return (node < n ? Xa : Xo) + (size_t)node * F;
  };
  // ---- fragments of the logit GEMM (rows past n*H clamped: those rows of L are cleared before they are used) ----
  float aq[RT][FQ], bx[CT][FQ];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    int row = rt * 16 + li;
    row = row < nH ? row : nH - 1;
    const float* p = qt + (size_t)row * F + lq * FQ;
#pragma unroll
    for (int u = 0; u < FQ; ++u) aq[rt][u] = p[u];
  }
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const float* p = xrow(ct * 16 + li) + lq * FQ;
#pragma unroll
    for (int u = 0; u < FQ; ++u) bx[ct][u] = p[u];
  }
  // ---- everything else the wave will need from global memory, issued behind the fragments: the x_i rows that are
  //      copied into zcat, and per softmax lane (8 lanes per (agent, head) pair, NP passes of 8 pairs) the mask and
  //      edge features of its SJ slots ----
  const int sub = lane & 7;
  float4 xcopy = make_float4(0.f, 0.f, 0.f, 0.f);
  if (lane < n * FQ) xcopy = reinterpret_cast<const float4*>(Xa)[lane];
  float mkv[NP][SJ];
  int pi[NP], ph[NP];
  {
    const float* mk = a.emask + (size_t)g * n * S;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      int pair = (lane >> 3) + 8 * p;
      pair = pair < nH ? pair : nH - 1;
      pi[p] = (pair * hmagic) >> 16;
      ph[p] = pair - pi[p] * H;
#pragma unroll
      for (int j = 0; j < SJ; ++j) {
        int sl = sub + 8 * j;
        sl = sl < S ? sl : S - 1;
        mkv[p][j] = mk[pi[p] * S + sl];
      }
    }
  }
  // node fragments of the aggregation GEMM and the edge features of this lane's slots.  Narrow layers (F = 8) have the
  // registers to request them together with the logit fragments — ONE global-memory round trip per graph; for F = 32
  // that would cost the third wave per SIMD (measured: 122 vs 113 us), so there they are requested after the logit GEMM,
  // when its fragments are dead, and land while the softmax runs.
  constexpr bool EARLY = (F <= 8);
  float bz[KZ][FT];
  float4 efv[NP][SJ];
  auto load_late_operands = [&]() {
#pragma unroll
    for (int k4 = 0; k4 < KZ; ++k4) {
      const float* p = xrow(k4 * 4 + lq);
#pragma unroll
      for (int ft = 0; ft < FT; ++ft) { const int col = ft * 16 + li; bz[k4][ft] = p[col < F ? col : F - 1]; }
    }
    const float* ef = a.efeat + (size_t)g * n * S * 4;
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int j = 0; j < SJ; ++j) {
        int sl = sub + 8 * j;
        sl = sl < S ? sl : S - 1;
        efv[p][j] = reinterpret_cast<const float4*>(ef)[pi[p] * S + sl];
      }
  };
  if constexpr (EARLY) { load_late_operands(); __builtin_amdgcn_sched_barrier(0); }
  ASTAMP(1);
  // ---- L = Qt Xs^T, one row tile at a time, its column tiles interleaved (independent accumulators) ----
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    f32x4g acc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[ct] = f32x4g{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < FQ; ++u)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[rt][u], bx[ct][u], acc[ct], 0, 0, 0);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) s_L[(rt * 16 + lq * 4 + r) * Ll + ct * 16 + li] = acc[ct][r];
  }
  ASTAMP(2);
  if constexpr (!EARLY) { __builtin_amdgcn_sched_barrier(0); load_late_operands(); }
  // the parts of zcat that are plain copies: x_i, the constant column, zero padding
  if (lane < n * FQ) *reinterpret_cast<float4*>(zc + (lane / FQ) * Kp + 4 * (lane % FQ)) = xcopy;
  for (int idx = lane + 64; idx < n * FQ; idx += 64)
    *reinterpret_cast<float4*>(zc + (idx / FQ) * Kp + 4 * (idx % FQ)) = reinterpret_cast<const float4*>(Xa)[idx];
  {
    const int wpad = Kp - kc;                                  // >= 1: the constant column, then zeros
    for (int i = lane; i < n; i += 64)
      for (int c = 0; c < wpad; ++c) zc[i * Kp + kc + c] = (c == 0) ? 1.0f : 0.0f;
  }
  ASTAMP(3);
  // ---- gather + softmax + edge aggregation + scatter of P ----
  {
    float* at = a.attn + (size_t)g * n * S * H;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int pair = (lane >> 3) + 8 * p;
      const bool live = pair < nH;
      const int i = pi[p], h = ph[p];
      float* Lrow = s_L + (live ? pair : 0) * Ll;
      float l[SJ];
      int nd[SJ];
      float mx = -INFINITY;
#pragma unroll
      for (int j = 0; j < SJ; ++j) {
        const int sl = sub + 8 * j;
        const bool ok = live && sl < S;
        nd[j] = ok ? sender_node(t, i, sl) : 0;
        const float lv = Lrow[nd[j]];
        l[j] = (ok && mkv[p][j] != 0.0f) ? lv : -INFINITY;
        mx = fmaxf(mx, l[j]);
      }
      mx = grp8_max(mx);
      float den = 0.0f;
#pragma unroll
      for (int j = 0; j < SJ; ++j) {
        const float ev = (l[j] == -INFINITY) ? 0.0f : __expf(l[j] - mx)   /* v_exp_f32: rel. error ~1e-7 on weights <= 1 */;
        l[j] = ev;
        den += ev;
      }
      den = grp8_sum(den);
      const float inv = (den > 0.0f) ? 1.0f / den : 0.0f;
      if (live) {
#pragma unroll
        for (int c = 0; c < CT * 2; ++c) Lrow[sub + 8 * c] = 0.0f;   // own row: logits -> zeros -> P
      }
      float z0 = 0.f, z1 = 0.f, z2 = 0.f, z3 = 0.f;
#pragma unroll
      for (int j = 0; j < SJ; ++j) {
        const int sl = sub + 8 * j;
        if (live && sl < S) {
          const float av = l[j] * inv;
          if (a.attn != nullptr) at[(i * S + sl) * H + h] = av;
          if (av != 0.0f) {   // masked slots may carry 5e5 / NaN edge features: skip, never multiply
            Lrow[nd[j]] = av;
            const float4 e = efv[p][j];
            z0 = fmaf(av, e.x, z0); z1 = fmaf(av, e.y, z1); z2 = fmaf(av, e.z, z2); z3 = fmaf(av, e.w, z3);
          }
        }
      }
      z0 = grp8_sum(z0); z1 = grp8_sum(z1); z2 = grp8_sum(z2); z3 = grp8_sum(z3);
      if (live && sub == 0) *reinterpret_cast<float4*>(zc + i * Kp + F + h * Wd + F) = make_float4(z0, z1, z2, z3);
    }
  }
  // rows of the last row tile past n*H still hold logits of clamped query rows: P must be zero there
  for (int idx = nH * Ll + lane; idx < RT * 16 * Ll; idx += 64) s_L[idx] = 0.0f;
  ASTAMP(4);
  // ---- Zx = P Xs: all (row tile, feature tile) accumulators interleaved over the node k-steps ----
  {
    f32x4g acc[RT][FT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int ft = 0; ft < FT; ++ft) acc[rt][ft] = f32x4g{0.f, 0.f, 0.f, 0.f};
    // columns of P past Ns are zero (cleared, never scattered to), so clamped node rows contribute nothing
#pragma unroll
    for (int k0 = 0; k0 < KZ; k0 += 8) {
      float pa[8][RT];
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
          if (k0 + u < KZ) pa[u][rt] = s_L[(rt * 16 + li) * Ll + 4 * (k0 + u) + lq];
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int ft = 0; ft < FT; ++ft)
            if (k0 + u < KZ) acc[rt][ft] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[u][rt], bz[k0 + u][ft], acc[rt][ft], 0, 0, 0);
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = rt * 16 + lq * 4 + r;
        const int i = (row * hmagic) >> 16, h = row - i * H;
#pragma unroll
        for (int ft = 0; ft < FT; ++ft) {
          const int col = ft * 16 + li;
          if (row < nH && col < F) zc[i * Kp + F + h * Wd + col] = acc[rt][ft][r];
        }
      }
  }
  ASTAMP(5); ASTAMP(6);
}

// ---- backward, one wave per graph ----------------------------------------------------------------------------------
// Same layout rules as attn_fwd_wave_kernel.  The wave's LDS tile [RT*16][Ll] is used three times in sequence:
//   1. dA = dZx Xs^T (dense), read back at the slots by the 8 lanes that own an (agent, head) pair, which compute the
//      softmax backward dl = a (dA + dze.e - sum a (dA + dze.e)) in registers;
//   2. the tile is overwritten with P (own rows, in place) and dXs += P^T dZx runs on the matrix cores;
//   3. the tile is overwritten with dL and dQt = dL Xs, dXs += dL^T Qt run.
// DS operations of a wave execute in order, so none of these hand-overs needs a barrier.  The dXs accumulators
// (CT x ceil(F/16) tiles) stay in registers across 2 and 3.
template <int F, int CT, int NP, int SJ>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DGPPO_ATTN_BWD_WPE, 8))) attn_bwd_wave_kernel(AttnArgs a) {
  extern __shared__ float sm[];
  constexpr int FQ = F / 4, FT = (F + 15) / 16, KZ = CT * 4, Ll = CT * 16 + 1, RT = (NP + 1) / 2, KP = RT * 4;
  const Topo& t = a.t;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lq = lane >> 4;
  // wave-uniform by construction; readfirstlane tells the compiler, so the per-graph base pointers live in SGPRs and the
  // loads use scalar-base + 32-bit-offset addressing instead of 64-bit vector address arithmetic
  const int g = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave);
  if (g >= a.G) return;
  const int n = t.n, S = t.S, Ns = t.Ns, H = a.H, nH = n * H, Kp = a.Kp;
  const int Wd = F + 4;
  const int hmagic = (65536 + H - 1) / H;
  float* s_T = sm + wave * (RT * 16 * Ll);
  const float* Xa = a.Xa + (size_t)g * n * F;
  const float* Xo = a.Xo + (size_t)g * (Ns - n) * F - (size_t)n * F;
  const float* qt = a.qt + (size_t)g * nH * F;
  const float* dzc = a.dzcat + (size_t)g * n * Kp;
  const bool want_dx = a.dXa != nullptr;
  auto xrow = [&](int node) -> const float* {
    node = node < Ns ? node : Ns - 1;
    return (node < n ? Xa : Xo) + (size_t)node * F;
  };
  auto dzrow = [&](int pair) -> const float* {                 // dZx row of an (agent, head) pair inside dzcat (clamped)
    pair = pair < nH ? pair : nH - 1;
    const int i = (pair * hmagic) >> 16, h = pair - i * H;
    return dzc + i * Kp + F + h * Wd;
  };
  // ---- fragments of dA = dZx Xs^T (k-permuted: lane (li, lq) takes features lq*F/4 .. of its row) ----
  float adz[RT][FQ], bx[CT][FQ];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const float* p = dzrow(rt * 16 + li) + lq * FQ;
#pragma unroll
    for (int u = 0; u < FQ; ++u) adz[rt][u] = p[u];
  }
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const float* p = xrow(ct * 16 + li) + lq * FQ;
#pragma unroll
    for (int u = 0; u < FQ; ++u) bx[ct][u] = p[u];
  }
  // ---- per softmax lane: attention weights, edge features and the edge part of dZ for its slots ----
  const int sub = lane & 7;
  float av[NP][SJ], ed[NP][SJ];
  int pi[NP], ph[NP];
  {
    const float* ef = a.efeat + (size_t)g * n * S * 4;
    const float* at = a.attn + (size_t)g * n * S * H;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      int pair = (lane >> 3) + 8 * p;
      const bool live = pair < nH;
      pair = live ? pair : nH - 1;
      pi[p] = (pair * hmagic) >> 16;
      ph[p] = pair - pi[p] * H;
      const float4 dze = *reinterpret_cast<const float4*>(dzc + pi[p] * Kp + F + ph[p] * Wd + F);
#pragma unroll
      for (int j = 0; j < SJ; ++j) {
        int sl = sub + 8 * j;
        const bool ok = live && sl < S;
        sl = sl < S ? sl : S - 1;
        const float w = at[(pi[p] * S + sl) * H + ph[p]];
        av[p][j] = ok ? w : 0.0f;
        const float4 e = reinterpret_cast<const float4*>(ef)[pi[p] * S + sl];
        // masked slots (a == 0) may carry 5e5 / NaN edge features: never multiply them
        ed[p][j] = (ok && w != 0.0f) ? fmaf(dze.x, e.x, fmaf(dze.y, e.y, fmaf(dze.z, e.z, dze.w * e.w))) : 0.0f;
      }
    }
  }
  // ---- dA (dense) -> LDS tile ----
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    f32x4g acc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[ct] = f32x4g{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < FQ; ++u)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(adz[rt][u], bx[ct][u], acc[ct], 0, 0, 0);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) s_T[(rt * 16 + lq * 4 + r) * Ll + ct * 16 + li] = acc[ct][r];
  }
  __builtin_amdgcn_sched_barrier(0);
  // ---- fragments needed later, requested now: dZx and Qt as B operands (k = pair), Xs as B operand (k = node) ----
  float bdz[KP][FT], bq[KP][FT], bz[KZ][FT];
#pragma unroll
  for (int k4 = 0; k4 < KP; ++k4) {
    const int pair = k4 * 4 + lq;
    const float* pz = dzrow(pair);
    const float* pq = qt + (size_t)(pair < nH ? pair : nH - 1) * F;
#pragma unroll
    for (int ft = 0; ft < FT; ++ft) {
      const int col = ft * 16 + li, cc = col < F ? col : F - 1;
      bdz[k4][ft] = want_dx ? pz[cc] : 0.0f;
      bq[k4][ft] = want_dx ? pq[cc] : 0.0f;
    }
  }
#pragma unroll
  for (int k4 = 0; k4 < KZ; ++k4) {
    const float* p = xrow(k4 * 4 + lq);
#pragma unroll
    for (int ft = 0; ft < FT; ++ft) { const int col = ft * 16 + li; bz[k4][ft] = p[col < F ? col : F - 1]; }
  }
  // ---- softmax backward in registers; the tile becomes P ----
  float dl[NP][SJ];
  int nd[NP][SJ];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int pair = (lane >> 3) + 8 * p;
    const bool live = pair < nH;
    float* Trow = s_T + (live ? pair : 0) * Ll;
    float dot = 0.0f;
#pragma unroll
    for (int j = 0; j < SJ; ++j) {
      const int sl = sub + 8 * j;
      nd[p][j] = (live && sl < S) ? sender_node(t, pi[p], sl) : 0;
      const float dA = (av[p][j] != 0.0f) ? Trow[nd[p][j]] + ed[p][j] : 0.0f;
      dl[p][j] = dA;
      dot = fmaf(av[p][j], dA, dot);
    }
    dot = grp8_sum(dot);
#pragma unroll
    for (int j = 0; j < SJ; ++j) dl[p][j] = av[p][j] * (dl[p][j] - dot);
    if (live) {
#pragma unroll
      for (int cc = 0; cc < CT * 2; ++cc) Trow[sub + 8 * cc] = 0.0f;
#pragma unroll
      for (int j = 0; j < SJ; ++j) if (av[p][j] != 0.0f) Trow[nd[p][j]] = av[p][j];
    }
  }
  // rows past n*H hold dA of clamped rows: they must be zero in every later use of the tile
  for (int idx = nH * Ll + lane; idx < RT * 16 * Ll; idx += 64) s_T[idx] = 0.0f;
  // ---- dXs += P^T dZx ----
  f32x4g dxacc[CT][FT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int ft = 0; ft < FT; ++ft) dxacc[ct][ft] = f32x4g{0.f, 0.f, 0.f, 0.f};
  if (want_dx) {
#pragma unroll
    for (int k4 = 0; k4 < KP; ++k4) {
      float pa[CT];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) pa[ct] = s_T[(k4 * 4 + lq) * Ll + ct * 16 + li];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int ft = 0; ft < FT; ++ft) dxacc[ct][ft] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[ct], bdz[k4][ft], dxacc[ct][ft], 0, 0, 0);
    }
  }
  // ---- the tile becomes dL ----
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int pair = (lane >> 3) + 8 * p;
    if (pair < nH) {
      float* Trow = s_T + pair * Ll;
#pragma unroll
      for (int cc = 0; cc < CT * 2; ++cc) Trow[sub + 8 * cc] = 0.0f;
#pragma unroll
      for (int j = 0; j < SJ; ++j) if (dl[p][j] != 0.0f) Trow[nd[p][j]] = dl[p][j];
    }
  }
  // ---- dQt = dL Xs ----
  {
    f32x4g acc[RT][FT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int ft = 0; ft < FT; ++ft) acc[rt][ft] = f32x4g{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k0 = 0; k0 < KZ; k0 += 8) {
      float pa[8][RT];
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
          if (k0 + u < KZ) pa[u][rt] = s_T[(rt * 16 + li) * Ll + 4 * (k0 + u) + lq];
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int ft = 0; ft < FT; ++ft)
            if (k0 + u < KZ) acc[rt][ft] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[u][rt], bz[k0 + u][ft], acc[rt][ft], 0, 0, 0);
    }
    float* dq = a.dqt + (size_t)g * nH * F;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = rt * 16 + lq * 4 + r;
#pragma unroll
        for (int ft = 0; ft < FT; ++ft) {
          const int col = ft * 16 + li;
          if (row < nH && col < F) dq[row * F + col] = acc[rt][ft][r];
        }
      }
  }
  // ---- dXs += dL^T Qt, then the stores (+ the direct x_i part of dzcat for agents) ----
  if (want_dx) {
#pragma unroll
    for (int k4 = 0; k4 < KP; ++k4) {
      float pa[CT];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) pa[ct] = s_T[(k4 * 4 + lq) * Ll + ct * 16 + li];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int ft = 0; ft < FT; ++ft) dxacc[ct][ft] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[ct], bq[k4][ft], dxacc[ct][ft], 0, 0, 0);
    }
    // Epilogue in two passes.  What every output element still needs from memory — the direct x_i part of dzcat for an agent
    // row, the sign of the node's own feature for the ReLU mask of another row — is READ FOR ALL ELEMENTS FIRST and the stores
    // follow.  Interleaved (load, use, store per element) the compiler must keep program order between a store and the next
    // load (the pointers may alias) and `s_waitcnt vmcnt(0)` before each use also waits for every earlier STORE: 40 fully
    // serialised memory round trips per graph, 60 % of the kernel's wave time (SQ_WAIT_ANY, profiles/r03_nn_counters.json).
    float side[CT][FT][4];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int node = ct * 16 + lq * 4 + r;
#pragma unroll
        for (int ft = 0; ft < FT; ++ft) {
          const int f = ft * 16 + li;
          float sv = 1.0f;
          if (node < Ns && f < F) {
            if (node < n) sv = dzc[node * Kp + f];
            else if (a.dXo != nullptr && a.relu_xo) sv = a.Xo[((size_t)g * (Ns - n) + (node - n)) * F + f];   // just read as a fragment: an L2 hit
          }
          side[ct][ft][r] = sv;
        }
      }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int node = ct * 16 + lq * 4 + r;
#pragma unroll
        for (int ft = 0; ft < FT; ++ft) {
          const int f = ft * 16 + li;
          if (node >= Ns || f >= F) continue;
          if (node < n) a.dXa[((size_t)g * n + node) * F + f] = dxacc[ct][ft][r] + side[ct][ft][r];
          else if (a.dXo != nullptr) {
            const size_t o = ((size_t)g * (Ns - n) + (node - n)) * F + f;
            float v = dxacc[ct][ft][r];
            if (a.relu_xo) v = (side[ct][ft][r] > 0.0f) ? v : 0.0f;
            a.dXo[o] = v;
          }
        }
      }
  }
}

// ---- F = 32, block-diagonal form (one wave per graph) -----------------------------------------------------------
// The dense wave kernels above form the whole [n*H x nodes] logit tile although an agent only attends to the nodes every
// agent sees (agents, goals: "shared") and to its OWN 8 LiDAR hit nodes: 4/5 of the tile's columns (LidarSpread n = 8: 64
// of 80) are used by one agent in eight.  Here the wave is laid out as 8 lanes per AGENT (il = lane / 8, sub = lane & 7):
//   * lane (il, sub) owns, for every head, the slots  {shared node p * 8 + sub, p < PS}  and  {hit sub}  of agent il, i.e.
//     all S slots of an agent live in its 8 lanes (PS + 1 registers per head) and the softmax is a register / DPP matter;
//   * all products run on the matrix cores as v_mfma_f32_4x4x1 (16 independent 4x4 blocks, one k per instruction): block
//     (il, g2 = bit 2 of the lane), A rows = the 4 heads of agent il, B columns = 4 nodes (logits, dA) or 4 feature quads
//     (aggregation, dQt).  A block only ever multiplies what its agent needs: 96 + 96 small MFMAs (1.5 k SIMD cycles) per
//     graph forward instead of 160 16x16x4 tiles (5.1 k), 2.7 k instead of 10.2 k backward;
//   * every global row is read ONCE, as coalesced 16-byte pieces, into a padded LDS image (shared rows, the 64 hit rows of
//     the agent batch, query / dZ rows) from which all operand layouts are read; the dense kernels re-read the node rows
//     from global memory per layout (2.4x the algorithmic bytes, profiles/r03_nn_counters.json).
// More than 8 agents: batches of 8 agents run one after the other over the same staged shared rows.
// Requirements (else the dense kernels above): F = 32, H <= 4, LiDAR hits 8 per agent (or no private nodes at all), at most
// 32 shared nodes.  Same arithmetic as gnn.py:85-117 up to the summation order.
#define ABD_XL 36                      // LDS row stride of a staged 32-float row (16-byte aligned, conflict-free b128 row reads)
#define ABD_DZL 40                     // the staged dZ rows carry the 4 edge-feature gradients behind the 32 features
typedef float4 abd_f4;
__device__ inline f32x4g mfma4(float a_, float b_, f32x4g c_) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a_, b_, c_, 0, 0, 0); }
__device__ inline const float* f4e(const float4& v) { return reinterpret_cast<const float*>(&v); }
__device__ inline float4 ld4_if(bool ok, const float4* p) {      // predicated 16-byte load, zeros otherwise
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (ok) v = *p;
  return v;
}

// An opaque zero that DEPENDS on loaded values (v_and_b32 with 0 behind an asm the optimiser cannot see through): added to the
// LDS addresses of the staging writes it ties the loads of the small operands (masks, edge features, attention weights ...)
// to the first trip to memory.  Left alone the scheduler sinks them into the first MFMA phase and the wave pays a second trip;
// a side-effecting fence (sched_barrier, asm volatile) at that place makes the compiler keep the staging arrays in scratch.
__device__ inline int opaque_zero(int bits) {
  asm("v_and_b32 %0, 0, %0" : "+v"(bits));
  return bits;
}
__device__ inline int f4bits(const float4& v) { return __float_as_int(v.x) | __float_as_int(v.y) | __float_as_int(v.z) | __float_as_int(v.w); }

template <int PS, bool HITS> struct AbdLds {
  static constexpr int NSP = PS * 8, PC = NSP + (HITS ? 8 : 0), PL = PC + 4;
  static constexpr int XS = NSP * ABD_XL, XH = HITS ? 64 * ABD_XL + 64 : 0;
  static constexpr int QP = (32 * ABD_XL > 32 * PL) ? 32 * ABD_XL : 32 * PL;      // query rows, later the P tile [32][PL]
  static constexpr int FWD = XS + XH + QP;
  static constexpr int DZ = 32 * ABD_DZL, PTL = NSP + 1, PT = 32 * PTL, DL = 32 * PL;
  static constexpr int BWD = XS + XH + DZ + PT + DL;
};

// ---- other nodes recomputed in the kernel (XOF variants) ----
// Goals, LiDAR hits and obstacles receive no messages, so their layer-1 features are relu(x W_u[:8] + b_u) of the 8 raw
// features (gnn.py:109-111 with aggr = 0).  Materialised, those rows are 9 KB of the 19 KB a graph's forward moves (and are
// written once and read again by the backward); here the wave reads the 32-byte raw rows and forms the 16-row x 32 tiles on
// the matrix cores (2 k-steps x 2 column tiles of 16x16x4 per 16 rows) straight into the LDS images.
#define ABD_KR 8
#define ABD_DW_STRIDE 320              // 8 x 32 (dWo) + 32 (dbo), rounded to 64 floats
template <int GT> struct AbdXoRegs { float ag[GT][2], ah[4][2], wo[2][2], bo[2]; };
// request the A fragments: row li of each 16-row tile, raw feature lq + 4 s
template <int GT, bool HITS>
__device__ inline void abd_xo_load(AbdXoRegs<GT>& x, const AttnArgs& a, const float* raw, int n_shared_other, int ng, int hit0,
                                   int n_hits, int li, int lq, bool first) {
  if (first) {
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) x.wo[s_][ct] = a.Wo[(lq + 4 * s_) * a.ldwo + ct * 16 + li];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) x.bo[ct] = a.bo[ct * 16 + li];
#pragma unroll
    for (int t = 0; t < GT; ++t) {
      int o = t * 16 + li;
      o = o < n_shared_other ? o : n_shared_other - 1;
      o = o < 0 ? 0 : o;
#pragma unroll
      for (int s_ = 0; s_ < 2; ++s_) x.ag[t][s_] = raw[o * ABD_KR + lq + 4 * s_];
    }
  }
  if constexpr (HITS) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      int hb = hit0 + t * 16 + li;
      hb = hb < n_hits ? hb : n_hits - 1;
#pragma unroll
      for (int s_ = 0; s_ < 2; ++s_) x.ah[t][s_] = raw[(ng + hb) * ABD_KR + lq + 4 * s_];
    }
  }
}
// tiles -> LDS images (C/D layout: row lq * 4 + r, column ct * 16 + li)
template <int GT, bool HITS>
__device__ inline void abd_xo_emit(const AbdXoRegs<GT>& x, float* XS, float* XH, int n, int n_shared_other, int li, int lq, bool first) {
  auto tile = [&](const float (&ar)[2], f32x4g (&v)[2]) {
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      v[ct] = f32x4g{0.f, 0.f, 0.f, 0.f};
      v[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[0], x.wo[0][ct], v[ct], 0, 0, 0);
      v[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[1], x.wo[1][ct], v[ct], 0, 0, 0);
    }
  };
  if (first) {
#pragma unroll
    for (int t = 0; t < GT; ++t) {
      f32x4g v[2];
      tile(x.ag[t], v);
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int o = t * 16 + lq * 4 + r;
          if (o < n_shared_other) XS[(n + o) * ABD_XL + ct * 16 + li] = fmaxf(v[ct][r] + x.bo[ct], 0.0f);
        }
    }
  }
  if constexpr (HITS) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      f32x4g v[2];
      tile(x.ah[t], v);
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int hr = t * 16 + lq * 4 + r;
          XH[hr * ABD_XL + (hr >> 3) * 8 + ct * 16 + li] = fmaxf(v[ct][r] + x.bo[ct], 0.0f);
        }
    }
  }
}
template <int GT> __device__ inline int abd_xo_bits(const AbdXoRegs<GT>& x, bool hits) {
  int b = 0;
#pragma unroll
  for (int t = 0; t < GT; ++t) b |= __float_as_int(x.ag[t][0]) | __float_as_int(x.ag[t][1]);
  if (hits) {
#pragma unroll
    for (int t = 0; t < 4; ++t) b |= __float_as_int(x.ah[t][0]) | __float_as_int(x.ah[t][1]);
  }
  b |= __float_as_int(x.wo[0][0]) | __float_as_int(x.wo[0][1]) | __float_as_int(x.wo[1][0]) | __float_as_int(x.wo[1][1]);
  b |= __float_as_int(x.bo[0]) | __float_as_int(x.bo[1]);
  return b;
}

template <int PS, bool HITS, int AB, bool XOF>
__global__ void __launch_bounds__(128) attn_fwd_bd_kernel(AttnArgs a) {
  extern __shared__ float4 abd_sm[];
  using L = AbdLds<PS, HITS>;
  constexpr int F = 32, NSP = L::NSP, PC = L::PC, PL = L::PL, NPR = PS + (HITS ? 1 : 0), Wd = F + 4;
  const Topo& t = a.t;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = __builtin_amdgcn_readfirstlane(blockIdx.x * 2 + wave);
  if (g >= a.G) return;                                        // no barriers below
  const int n = t.n, S = t.S, Ns = t.Ns, H = a.H, Kp = a.Kp;
  const int NSH = HITS ? t.n + t.ng : Ns;
  const int kc = F + H * Wd;
  float* XS = reinterpret_cast<float*>(abd_sm) + wave * L::FWD;
  float* XH = XS + L::XS;
  float* QP = XH + L::XH;
  const float4* Xa4 = reinterpret_cast<const float4*>(a.Xa + (size_t)g * n * F);
  const float4* Xo4 = reinterpret_cast<const float4*>(a.Xo + (size_t)g * (Ns - n) * F);
  const float4* qt4 = reinterpret_cast<const float4*>(a.qt + (size_t)g * n * H * F);
  float* zc = a.zcat + (size_t)g * n * Kp;
  const int il = lane >> 3, sub = lane & 7, g2 = (lane >> 2) & 1, c = lane & 3, hA = lane & 3;
  ASTAMP(0);
  // ---- every global read is requested before anything waits: rows are fetched unconditionally from clamped addresses (a
  //      pad row duplicates a real one; it only ever meets zeros of P), so there is no branch around a load and the wave
  //      makes ONE trip to memory per agent batch ----
  float4 vs[PS];
#pragma unroll
  for (int p = 0; p < PS; ++p) {
    const int row = p * 8 + il;
    const int ra = row < n ? row : n - 1, ro = (row < NSH ? row : NSH - 1) - n;
    const float4* src = (XOF || row < n || NSH == n) ? Xa4 + ra * 8 + sub : Xo4 + ro * 8 + sub;
    vs[p] = *src;
  }
  constexpr int GT = (NSP + 15) / 16;
  AbdXoRegs<GT> xo;
  const float* raw = XOF ? a.Xo_raw + (size_t)g * (Ns - n) * ABD_KR : nullptr;
  const int li_ = lane & 15, lq_ = lane >> 4;
  const float* mk = a.emask + (size_t)g * n * S;
  const float4* ef4 = reinterpret_cast<const float4*>(a.efeat + (size_t)g * n * S * 4);
  float* at = a.attn + (size_t)g * n * S * H;
#pragma unroll
  for (int ab = 0; ab < AB; ++ab) {       // AB = ceil(n / 8), a compile-time count: straight-line code, every array in registers
    const int i = ab * 8 + il;
    const bool live = i < n;
    const int ic = live ? i : n - 1;
    float4 vh[HITS ? 8 : 1], vq[4];
    if constexpr (XOF) abd_xo_load<GT, HITS>(xo, a, raw, NSH - n, t.ng, ab * 64, n * 8, li_, lq_, ab == 0);
    if constexpr (HITS && !XOF) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        int hb = ab * 64 + k * 8 + il;
        hb = hb < n * 8 ? hb : n * 8 - 1;
        vh[k] = Xo4[(t.ng + hb) * 8 + sub];
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int row = k * 8 + il;
      int ri = ab * 8 + (row >> 2), rh = row & 3;
      ri = ri < n ? ri : n - 1; rh = rh < H ? rh : H - 1;
      vq[k] = qt4[(ri * H + rh) * 8 + sub];
    }
    // the lane's slots: mask and edge features
    int slot[NPR];
    float mkv[NPR];
    float4 efv[NPR];
#pragma unroll
    for (int p = 0; p < NPR; ++p) {
      int sl;
      if (HITS && p == PS) sl = n + t.gs + sub;
      else { const int j = p * 8 + sub; sl = (j < NSH) ? slot_of(t, j, ic) : -1; }
      slot[p] = live ? sl : -1;
      const int sc = sl < 0 ? 0 : sl;
      mkv[p] = mk[ic * S + sc];
      efv[p] = ef4[ic * S + sc];
    }
    ASTAMP(1);
    int pin = 0;
#pragma unroll
    for (int p = 0; p < NPR; ++p) pin |= __float_as_int(mkv[p]) | f4bits(efv[p]);
    if constexpr (XOF) pin |= abd_xo_bits<GT>(xo, HITS);
    const int pin0 = opaque_zero(pin);
    if (ab == 0) {
#pragma unroll
      for (int p = 0; p < PS; ++p) {
        float4 v = vs[p];
        if (XOF && p * 8 + il >= n) v = make_float4(0.f, 0.f, 0.f, 0.f);       // pad rows stay zero, the others are recomputed below
        *reinterpret_cast<float4*>(XS + pin0 + (p * 8 + il) * ABD_XL + sub * 4) = v;
      }
    }
    if constexpr (XOF) abd_xo_emit<GT, HITS>(xo, XS + pin0, XH + pin0, n, NSH - n, li_, lq_, ab == 0);
    if constexpr (HITS && !XOF) {
#pragma unroll
      for (int k = 0; k < 8; ++k) *reinterpret_cast<float4*>(XH + pin0 + (k * 8 + il) * ABD_XL + k * 8 + sub * 4) = vh[k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) *reinterpret_cast<float4*>(QP + pin0 + (k * 8 + il) * ABD_XL + sub * 4) = vq[k];
#pragma unroll
    for (int p = 0; p < NPR; ++p) mkv[p] = (slot[p] >= 0) ? mkv[p] : 0.0f;
    ASTAMP(2);
    // ---- logits: A = the query row of (agent il, head hA), B = the lane's own node rows ----
    f32x4g acc[NPR];
#pragma unroll
    for (int p = 0; p < NPR; ++p) acc[p] = f32x4g{0.f, 0.f, 0.f, 0.f};
    {
      const float* qrow = QP + (il * 4 + hA) * ABD_XL;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float4 q = *reinterpret_cast<const float4*>(qrow + k * 4);
        float4 x[NPR];
#pragma unroll
        for (int p = 0; p < NPR; ++p) {
          if (HITS && p == PS) x[p] = *reinterpret_cast<const float4*>(XH + lane * ABD_XL + il * 8 + k * 4);
          else x[p] = *reinterpret_cast<const float4*>(XS + (p * 8 + sub) * ABD_XL + k * 4);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int p = 0; p < NPR; ++p) acc[p] = mfma4(f4e(q)[u], f4e(x[p])[u], acc[p]);
      }
    }
    ASTAMP(3);
    // ---- masked softmax per head over the 8 lanes x NPR registers of the agent; edge aggregation; P -> LDS ----
    float* PT = QP;                                            // DS operations of a wave execute in order: the query rows are read
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      if (h < H) {
        float l[NPR];
        float mx = -INFINITY;
#pragma unroll
        for (int p = 0; p < NPR; ++p) { l[p] = (mkv[p] != 0.0f) ? acc[p][h] : -INFINITY; mx = fmaxf(mx, l[p]); }
        mx = grp8_max(mx);
        float den = 0.0f;
#pragma unroll
        for (int p = 0; p < NPR; ++p) { l[p] = (l[p] == -INFINITY) ? 0.0f : __expf(l[p] - mx); den += l[p]; }
        den = grp8_sum(den);
        const float inv = (den > 0.0f) ? 1.0f / den : 0.0f;
        float z0 = 0.f, z1 = 0.f, z2 = 0.f, z3 = 0.f;
#pragma unroll
        for (int p = 0; p < NPR; ++p) {
          const float av = l[p] * inv;
          acc[p][h] = av;
          if (slot[p] >= 0) {
            if (a.attn != nullptr) at[(i * S + slot[p]) * H + h] = av;
            if (av != 0.0f) {   // masked slots may carry 5e5 / NaN edge features: skip, never multiply
              const float4 e = efv[p];
              z0 = fmaf(av, e.x, z0); z1 = fmaf(av, e.y, z1); z2 = fmaf(av, e.z, z2); z3 = fmaf(av, e.w, z3);
            }
          }
        }
        z0 = grp8_sum(z0); z1 = grp8_sum(z1); z2 = grp8_sum(z2); z3 = grp8_sum(z3);
        if (live && sub == 0) *reinterpret_cast<float4*>(zc + i * Kp + F + h * Wd + F) = make_float4(z0, z1, z2, z3);
      } else {
#pragma unroll
        for (int p = 0; p < NPR; ++p) acc[p][h] = 0.0f;
      }
#pragma unroll
      for (int p = 0; p < NPR; ++p) PT[(il * 4 + h) * PL + p * 8 + sub] = acc[p][h];
    }
    ASTAMP(4);
    // ---- Zx = P Xs: A = row (il, hA) of P, B = feature quad (g2 * 4 + c) of the node of the k-step ----
    {
      float pa[PC];
#pragma unroll
      for (int k = 0; k < PC / 4; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(PT + (il * 4 + hA) * PL + k * 4);
        pa[k * 4] = v.x; pa[k * 4 + 1] = v.y; pa[k * 4 + 2] = v.z; pa[k * 4 + 3] = v.w;
      }
      f32x4g az[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) az[q] = f32x4g{0.f, 0.f, 0.f, 0.f};
      const int fq = (g2 * 4 + c) * 4;
#pragma unroll
      for (int kk = 0; kk < NSP; ++kk) {
        const float4 b = *reinterpret_cast<const float4*>(XS + kk * ABD_XL + fq);
#pragma unroll
        for (int q = 0; q < 4; ++q) az[q] = mfma4(pa[kk], f4e(b)[q], az[q]);
      }
      if constexpr (HITS) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          const float4 b = *reinterpret_cast<const float4*>(XH + (il * 8 + m) * ABD_XL + il * 8 + fq);
#pragma unroll
          for (int q = 0; q < 4; ++q) az[q] = mfma4(pa[NSP + m], f4e(b)[q], az[q]);
        }
      }
      if (live) {
#pragma unroll
        for (int h = 0; h < 4; ++h)
          if (h < H) *reinterpret_cast<float4*>(zc + i * Kp + F + h * Wd + fq) = make_float4(az[0][h], az[1][h], az[2][h], az[3][h]);
      }
    }
  }
  ASTAMP(5);
  // the parts of zcat that are plain copies: x_i (from its staged image), the constant column, zero padding
#pragma unroll
  for (int p = 0; p < PS; ++p) {
    const int row = p * 8 + il;
    if (row < n) *reinterpret_cast<float4*>(zc + row * Kp + sub * 4) = *reinterpret_cast<const float4*>(XS + row * ABD_XL + sub * 4);
  }
  {
    const int wpad = Kp - kc;                                  // >= 1: the constant column, then zeros
    for (int i = lane; i < n; i += 64)
      for (int cc = 0; cc < wpad; ++cc) zc[i * Kp + kc + cc] = (cc == 0) ? 1.0f : 0.0f;
  }
  ASTAMP(6);
}

// ---- the same forward with PERSISTENT waves and the next graph's loads in flight (n <= 8: one agent batch) ----------------
// A third of a wave's time per graph is the wait for its one trip to memory (in-kernel stamps: 6.7 k of 21 k cycles) and LDS
// limits a CU to 10 of these waves, so that wait is not hidden.  Here a wave walks graphs g, g + W, g + 2 W, ...; the loads of
// the next graph are requested right after the current one's rows are staged and land during its MFMA / softmax phases (+55
// registers: 8 instead of 10 waves per CU).  Launches of up to twice the resident waves keep the kernel above (rollouts).
template <int PS, bool HITS, bool XOF>
struct AbdFwdRegs {
  float4 vs[PS], vq[4], vh[(HITS && !XOF) ? 8 : 1];
  AbdXoRegs<(PS * 8 + 15) / 16> xo;
  float mkv[PS + (HITS ? 1 : 0)];
  float4 efv[PS + (HITS ? 1 : 0)];
};
template <int PS, bool HITS, bool XOF>
__global__ void __launch_bounds__(128) attn_fwd_bdp_kernel(AttnArgs a) {
  extern __shared__ float4 abd_sm[];
  using L = AbdLds<PS, HITS>;
  constexpr int F = 32, NSP = L::NSP, PC = L::PC, PL = L::PL, NPR = PS + (HITS ? 1 : 0), Wd = F + 4, GT = (NSP + 15) / 16;
  const Topo& t = a.t;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int g = __builtin_amdgcn_readfirstlane(blockIdx.x * 2 + wave);
  const int stride = gridDim.x * 2;
  if (g >= a.G) return;                                        // no barriers below
  const int n = t.n, S = t.S, Ns = t.Ns, H = a.H, Kp = a.Kp;
  const int NSH = HITS ? t.n + t.ng : Ns;
  const int kc = F + H * Wd;
  float* XS = reinterpret_cast<float*>(abd_sm) + wave * L::FWD;
  float* XH = XS + L::XS;
  float* QP = XH + L::XH;
  const int il = lane >> 3, sub = lane & 7, g2 = (lane >> 2) & 1, c = lane & 3, hA = lane & 3;
  const int li_ = lane & 15, lq_ = lane >> 4;
  const int i = il;
  const bool live = i < n;
  const int ic = live ? i : n - 1;
  // the lane's slots do not depend on the graph
  int slot[NPR], sc[NPR];
#pragma unroll
  for (int p = 0; p < NPR; ++p) {
    int sl;
    if (HITS && p == PS) sl = n + t.gs + sub;
    else { const int j = p * 8 + sub; sl = (j < NSH) ? slot_of(t, j, ic) : -1; }
    slot[p] = live ? sl : -1;
    sc[p] = sl < 0 ? 0 : sl;
  }
  using Regs = AbdFwdRegs<PS, HITS, XOF>;
  auto load = [&](int gg, Regs& R) {
    const float4* Xa4 = reinterpret_cast<const float4*>(a.Xa + (size_t)gg * n * F);
    const float4* Xo4 = reinterpret_cast<const float4*>(a.Xo + (size_t)gg * (Ns - n) * F);
    const float4* qt4 = reinterpret_cast<const float4*>(a.qt + (size_t)gg * n * H * F);
#pragma unroll
    for (int p = 0; p < PS; ++p) {
      const int row = p * 8 + il;
      const int ra = row < n ? row : n - 1, ro = (row < NSH ? row : NSH - 1) - n;
      const float4* src = (XOF || row < n || NSH == n) ? Xa4 + ra * 8 + sub : Xo4 + ro * 8 + sub;
      R.vs[p] = *src;
    }
    if constexpr (XOF) abd_xo_load<GT, HITS>(R.xo, a, a.Xo_raw + (size_t)gg * (Ns - n) * ABD_KR, NSH - n, t.ng, 0, n * 8, li_, lq_, true);
    if constexpr (HITS && !XOF) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        int hb = k * 8 + il;
        hb = hb < n * 8 ? hb : n * 8 - 1;
        R.vh[k] = Xo4[(t.ng + hb) * 8 + sub];
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int row = k * 8 + il;
      int ri = row >> 2, rh = row & 3;
      ri = ri < n ? ri : n - 1; rh = rh < H ? rh : H - 1;
      R.vq[k] = qt4[(ri * H + rh) * 8 + sub];
    }
    const float* mk = a.emask + (size_t)gg * n * S;
    const float4* ef4 = reinterpret_cast<const float4*>(a.efeat + (size_t)gg * n * S * 4);
#pragma unroll
    for (int p = 0; p < NPR; ++p) {
      R.mkv[p] = mk[ic * S + sc[p]];
      R.efv[p] = ef4[ic * S + sc[p]];
    }
  };
  auto stage = [&](Regs& R) {
    int pin = 0;
#pragma unroll
    for (int p = 0; p < NPR; ++p) pin |= __float_as_int(R.mkv[p]) | f4bits(R.efv[p]);
    if constexpr (XOF) pin |= abd_xo_bits<GT>(R.xo, HITS);
    const int pin0 = opaque_zero(pin);
#pragma unroll
    for (int p = 0; p < PS; ++p) {
      float4 v = R.vs[p];
      if (XOF && p * 8 + il >= n) v = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(XS + pin0 + (p * 8 + il) * ABD_XL + sub * 4) = v;
    }
    if constexpr (XOF) abd_xo_emit<GT, HITS>(R.xo, XS + pin0, XH + pin0, n, NSH - n, li_, lq_, true);
    if constexpr (HITS && !XOF) {
#pragma unroll
      for (int k = 0; k < 8; ++k) *reinterpret_cast<float4*>(XH + pin0 + (k * 8 + il) * ABD_XL + k * 8 + sub * 4) = R.vh[k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) *reinterpret_cast<float4*>(QP + pin0 + (k * 8 + il) * ABD_XL + sub * 4) = R.vq[k];
#pragma unroll
    for (int p = 0; p < NPR; ++p) R.mkv[p] = (slot[p] >= 0) ? R.mkv[p] : 0.0f;
  };
  auto compute = [&](int gg, const Regs& R) {
    float* zc = a.zcat + (size_t)gg * n * Kp;
    float* at = a.attn + (size_t)gg * n * S * H;
    f32x4g acc[NPR];
#pragma unroll
    for (int p = 0; p < NPR; ++p) acc[p] = f32x4g{0.f, 0.f, 0.f, 0.f};
    {
      const float* qrow = QP + (il * 4 + hA) * ABD_XL;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float4 q = *reinterpret_cast<const float4*>(qrow + k * 4);
        float4 x[NPR];
#pragma unroll
        for (int p = 0; p < NPR; ++p) {
          if (HITS && p == PS) x[p] = *reinterpret_cast<const float4*>(XH + lane * ABD_XL + il * 8 + k * 4);
          else x[p] = *reinterpret_cast<const float4*>(XS + (p * 8 + sub) * ABD_XL + k * 4);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int p = 0; p < NPR; ++p) acc[p] = mfma4(f4e(q)[u], f4e(x[p])[u], acc[p]);
      }
    }
    float* PT = QP;
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      if (h < H) {
        float l[NPR];
        float mx = -INFINITY;
#pragma unroll
        for (int p = 0; p < NPR; ++p) { l[p] = (R.mkv[p] != 0.0f) ? acc[p][h] : -INFINITY; mx = fmaxf(mx, l[p]); }
        mx = grp8_max(mx);
        float den = 0.0f;
#pragma unroll
        for (int p = 0; p < NPR; ++p) { l[p] = (l[p] == -INFINITY) ? 0.0f : __expf(l[p] - mx); den += l[p]; }
        den = grp8_sum(den);
        const float inv = (den > 0.0f) ? 1.0f / den : 0.0f;
        float z0 = 0.f, z1 = 0.f, z2 = 0.f, z3 = 0.f;
#pragma unroll
        for (int p = 0; p < NPR; ++p) {
          const float av = l[p] * inv;
          acc[p][h] = av;
          if (slot[p] >= 0) {
            if (a.attn != nullptr) at[(i * S + slot[p]) * H + h] = av;
            if (av != 0.0f) {   // masked slots may carry 5e5 / NaN edge features: skip, never multiply
              const float4 e = R.efv[p];
              z0 = fmaf(av, e.x, z0); z1 = fmaf(av, e.y, z1); z2 = fmaf(av, e.z, z2); z3 = fmaf(av, e.w, z3);
            }
          }
        }
        z0 = grp8_sum(z0); z1 = grp8_sum(z1); z2 = grp8_sum(z2); z3 = grp8_sum(z3);
        if (live && sub == 0) *reinterpret_cast<float4*>(zc + i * Kp + F + h * Wd + F) = make_float4(z0, z1, z2, z3);
      } else {
#pragma unroll
        for (int p = 0; p < NPR; ++p) acc[p][h] = 0.0f;
      }
#pragma unroll
      for (int p = 0; p < NPR; ++p) PT[(il * 4 + h) * PL + p * 8 + sub] = acc[p][h];
    }
    {
      float pa[PC];
#pragma unroll
      for (int k = 0; k < PC / 4; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(PT + (il * 4 + hA) * PL + k * 4);
        pa[k * 4] = v.x; pa[k * 4 + 1] = v.y; pa[k * 4 + 2] = v.z; pa[k * 4 + 3] = v.w;
      }
      f32x4g az[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) az[q] = f32x4g{0.f, 0.f, 0.f, 0.f};
      const int fq = (g2 * 4 + c) * 4;
#pragma unroll
      for (int kk = 0; kk < NSP; ++kk) {
        const float4 b = *reinterpret_cast<const float4*>(XS + kk * ABD_XL + fq);
#pragma unroll
        for (int q = 0; q < 4; ++q) az[q] = mfma4(pa[kk], f4e(b)[q], az[q]);
      }
      if constexpr (HITS) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          const float4 b = *reinterpret_cast<const float4*>(XH + (il * 8 + m) * ABD_XL + il * 8 + fq);
#pragma unroll
          for (int q = 0; q < 4; ++q) az[q] = mfma4(pa[NSP + m], f4e(b)[q], az[q]);
        }
      }
      if (live) {
#pragma unroll
        for (int h = 0; h < 4; ++h)
          if (h < H) *reinterpret_cast<float4*>(zc + i * Kp + F + h * Wd + fq) = make_float4(az[0][h], az[1][h], az[2][h], az[3][h]);
      }
    }
#pragma unroll
    for (int p = 0; p < PS; ++p) {
      const int row = p * 8 + il;
      if (row < n) *reinterpret_cast<float4*>(zc + row * Kp + sub * 4) = *reinterpret_cast<const float4*>(XS + row * ABD_XL + sub * 4);
    }
    {
      const int wpad = Kp - kc;                                  // >= 1: the constant column, then zeros
      for (int ii = lane; ii < n; ii += 64)
        for (int cc = 0; cc < wpad; ++cc) zc[ii * Kp + kc + cc] = (cc == 0) ? 1.0f : 0.0f;
    }
  };
  Regs cur;
  load(g, cur);
  for (;;) {
    stage(cur);
    const int gn = g + stride;
    const bool more = gn < a.G;                                // wave-uniform
    Regs nxt;
    if (more) load(gn, nxt);
    compute(g, cur);
    if (!more) break;
    cur = nxt;
    g = gn;
  }
}

// backward of the block-diagonal form: dA (as the logits), softmax backward in registers, dQt = dL Xs (as the aggregation),
// dXs of the shared nodes on 16x16x4 tiles (contraction over the batch's (agent, head) rows, accumulated over the agent
// batches), dXs of the hit nodes as 4x4 blocks whose A operand (P and dL of the lane's own hit) never leaves the registers.
template <int PS, bool HITS, int AB, bool XOF>
__global__ void __launch_bounds__(64) attn_bwd_bd_kernel(AttnArgs a) {
  extern __shared__ float4 abd_sm[];
  using L = AbdLds<PS, HITS>;
  constexpr int F = 32, NSP = L::NSP, PC = L::PC, PL = L::PL, PTL = L::PTL, NPR = PS + (HITS ? 1 : 0), Wd = F + 4;
  constexpr int RTS = (NSP + 15) / 16;
  const Topo& t = a.t;
  const int lane = threadIdx.x & 63;
  const int g = blockIdx.x;
  if (g >= a.G) return;
  const int n = t.n, S = t.S, Ns = t.Ns, H = a.H, Kp = a.Kp;
  const int NSH = HITS ? t.n + t.ng : Ns;
  float* XS = reinterpret_cast<float*>(abd_sm);
  float* XH = XS + L::XS;
  float* DZ = XH + L::XH;
  float* PT = DZ + L::DZ;
  float* DL = PT + L::PT;
  const float4* Xa4 = reinterpret_cast<const float4*>(a.Xa + (size_t)g * n * F);
  const float4* Xo4 = reinterpret_cast<const float4*>(a.Xo + (size_t)g * (Ns - n) * F);
  const float* qt = a.qt + (size_t)g * n * H * F;
  const float* dzc = a.dzcat + (size_t)g * n * Kp;
  const bool want_dx = a.dXa != nullptr;
  const int il = lane >> 3, sub = lane & 7, g2 = (lane >> 2) & 1, c = lane & 3, hA = lane & 3;
  const int li = lane & 15, lq = lane >> 4;
  // ---- every global read is requested before anything waits (clamped addresses, no branches: see the forward) ----
  float4 vs[PS];
#pragma unroll
  for (int p = 0; p < PS; ++p) {
    const int row = p * 8 + il;
    const int ra = row < n ? row : n - 1, ro = (row < NSH ? row : NSH - 1) - n;
    const float4* src = (XOF || row < n || NSH == n) ? Xa4 + ra * 8 + sub : Xo4 + ro * 8 + sub;
    vs[p] = *src;
  }
  constexpr int GT = (NSP + 15) / 16;
  AbdXoRegs<GT> xo;
  const float* raw = XOF ? a.Xo_raw + (size_t)g * (Ns - n) * ABD_KR : nullptr;
  const int li_ = lane & 15, lq_ = lane >> 4;
  // the direct x_i part of dzcat for the agent rows of dXa, in the C/D layout of the shared-node tiles
  float dir[RTS][2][4];
#pragma unroll
  for (int rt = 0; rt < RTS; ++rt)
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int node = rt * 16 + lq * 4 + r;
        dir[rt][ft][r] = dzc[(node < n ? node : n - 1) * Kp + ft * 16 + li];
      }
  f32x4g dxs[RTS][2];
#pragma unroll
  for (int rt = 0; rt < RTS; ++rt)
#pragma unroll
    for (int ft = 0; ft < 2; ++ft) dxs[rt][ft] = f32x4g{0.f, 0.f, 0.f, 0.f};
  f32x4g dwacc[2] = {f32x4g{0.f, 0.f, 0.f, 0.f}, f32x4g{0.f, 0.f, 0.f, 0.f}};      // XOF + dwo_slab: (dWo | dbo)^T, rows f, columns k
  float rbg[NSP / 4];
  const float4* ef4 = reinterpret_cast<const float4*>(a.efeat + (size_t)g * n * S * 4);
  const float* at = a.attn + (size_t)g * n * S * H;
#pragma unroll
  for (int ab = 0; ab < AB; ++ab) {       // AB = ceil(n / 8), a compile-time count: straight-line code, every array in registers
    const int i = ab * 8 + il;
    const bool live = i < n;
    const int ic = live ? i : n - 1;
    float4 vh[HITS ? 8 : 1], vz[5];
    if constexpr (XOF) abd_xo_load<GT, HITS>(xo, a, raw, NSH - n, t.ng, ab * 64, n * 8, li_, lq_, ab == 0);
    // B operands of the weight-gradient contraction (dgppo_attn_bwd_xo_dw): raw feature li & 7 of other-row 4 s + lq, requested
    // with everything else (read in the loop that uses them they were 16 dependent trips to L2)
    float rbh[HITS ? 16 : 1];
    if constexpr (XOF && HITS) {
#pragma unroll
      for (int s_ = 0; s_ < 16; ++s_) {
        int hb = ab * 64 + s_ * 4 + lq;
        hb = hb < n * 8 ? hb : n * 8 - 1;
        rbh[s_] = raw[(t.ng + hb) * ABD_KR + (li & 7)];
      }
    }
    if constexpr (XOF) {
      if (ab == 0) {
#pragma unroll
        for (int s_ = 0; s_ < NSP / 4; ++s_) {
          int o = s_ * 4 + lq;
          o = o < NSH - n ? o : NSH - n - 1;
          o = o < 0 ? 0 : o;
          rbg[s_] = raw[o * ABD_KR + (li & 7)];
        }
      }
    }
    if constexpr (HITS && !XOF) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        int hb = ab * 64 + k * 8 + il;
        hb = hb < n * 8 ? hb : n * 8 - 1;
        vh[k] = Xo4[(t.ng + hb) * 8 + sub];
      }
    }
    // dZ rows of the batch: 32 rows (agent, head) x 9 float4 (32 features + 4 edge terms); rows past n / H duplicate a real
    // row: they only meet zeros of P and dL
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      int tt = k * 64 + lane;
      tt = tt < 288 ? tt : 287;
      const int row = tt / 9, ch = tt - row * 9;
      int ri = ab * 8 + (row >> 2), rh = row & 3;
      ri = ri < n ? ri : n - 1; rh = rh < H ? rh : H - 1;
      vz[k] = *reinterpret_cast<const float4*>(dzc + ri * Kp + F + rh * Wd + ch * 4);
    }
    int slot[NPR];
    float4 efv[NPR];
    float av[NPR][4];
#pragma unroll
    for (int p = 0; p < NPR; ++p) {
      int sl;
      if (HITS && p == PS) sl = n + t.gs + sub;
      else { const int j = p * 8 + sub; sl = (j < NSH) ? slot_of(t, j, ic) : -1; }
      slot[p] = live ? sl : -1;
      const int sc = sl < 0 ? 0 : sl;
      efv[p] = ef4[ic * S + sc];
#pragma unroll
      for (int h = 0; h < 4; ++h) av[p][h] = at[(ic * S + sc) * H + (h < H ? h : H - 1)];
    }
    // operands that come straight from global memory: the query rows as B operands of the hit-node blocks (feature octet c
    // of every head of the lane's agent) and of the shared-node tiles (row = agent ks of the batch, head lq)
    float4 q8[4][2];
    if constexpr (HITS) {
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const float4* qr = reinterpret_cast<const float4*>(qt + (size_t)(ic * H + (h < H ? h : H - 1)) * F + c * 8);
        q8[h][0] = qr[0]; q8[h][1] = qr[1];
      }
    }
    float bQ[8][2];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int ri = ab * 8 + ks;
      const float* qr = qt + (size_t)((ri < n ? ri : n - 1) * H + (lq < H ? lq : H - 1)) * F;   // rows past n / H: dL is 0 there
#pragma unroll
      for (int ft = 0; ft < 2; ++ft) bQ[ks][ft] = qr[ft * 16 + li];
    }
    int pin = 0;
#pragma unroll
    for (int p = 0; p < NPR; ++p) {
      pin |= f4bits(efv[p]);
#pragma unroll
      for (int h = 0; h < 4; ++h) pin |= __float_as_int(av[p][h]);
    }
    if constexpr (HITS) {
#pragma unroll
      for (int h = 0; h < 4; ++h) pin |= f4bits(q8[h][0]) | f4bits(q8[h][1]);
    }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) pin |= __float_as_int(bQ[ks][0]) | __float_as_int(bQ[ks][1]);
    if (ab == 0) {
#pragma unroll
      for (int rt = 0; rt < RTS; ++rt)
#pragma unroll
        for (int ft = 0; ft < 2; ++ft)
#pragma unroll
          for (int r = 0; r < 4; ++r) pin |= __float_as_int(dir[rt][ft][r]);
    }
    if constexpr (XOF) pin |= abd_xo_bits<GT>(xo, HITS);
    if constexpr (XOF && HITS) {
#pragma unroll
      for (int s_ = 0; s_ < 16; ++s_) pin |= __float_as_int(rbh[s_]);
    }
    if constexpr (XOF) {
      if (ab == 0) {
#pragma unroll
        for (int s_ = 0; s_ < NSP / 4; ++s_) pin |= __float_as_int(rbg[s_]);
      }
    }
    const int pin0 = opaque_zero(pin);
    if (ab == 0) {
#pragma unroll
      for (int p = 0; p < PS; ++p) {
        float4 v = vs[p];
        if (XOF && p * 8 + il >= n) v = make_float4(0.f, 0.f, 0.f, 0.f);       // pad rows stay zero, the others are recomputed below
        *reinterpret_cast<float4*>(XS + pin0 + (p * 8 + il) * ABD_XL + sub * 4) = v;
      }
    }
    if constexpr (XOF) abd_xo_emit<GT, HITS>(xo, XS + pin0, XH + pin0, n, NSH - n, li_, lq_, ab == 0);
    if constexpr (HITS && !XOF) {
#pragma unroll
      for (int k = 0; k < 8; ++k) *reinterpret_cast<float4*>(XH + pin0 + (k * 8 + il) * ABD_XL + k * 8 + sub * 4) = vh[k];
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int tt = k * 64 + lane, row = tt / 9, ch = tt - row * 9;
      if (tt < 288) *reinterpret_cast<float4*>(DZ + pin0 + row * ABD_DZL + ch * 4) = vz[k];
    }
#pragma unroll
    for (int p = 0; p < NPR; ++p)
#pragma unroll
      for (int h = 0; h < 4; ++h) av[p][h] = (slot[p] >= 0 && h < H) ? av[p][h] : 0.0f;
    // ---- dA = dZx Xs^T at the lane's slots ----
    f32x4g acc[NPR];
#pragma unroll
    for (int p = 0; p < NPR; ++p) acc[p] = f32x4g{0.f, 0.f, 0.f, 0.f};
    {
      const float* zrow = DZ + (il * 4 + hA) * ABD_DZL;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float4 q = *reinterpret_cast<const float4*>(zrow + k * 4);
        float4 x[NPR];
#pragma unroll
        for (int p = 0; p < NPR; ++p) {
          if (HITS && p == PS) x[p] = *reinterpret_cast<const float4*>(XH + lane * ABD_XL + il * 8 + k * 4);
          else x[p] = *reinterpret_cast<const float4*>(XS + (p * 8 + sub) * ABD_XL + k * 4);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int p = 0; p < NPR; ++p) acc[p] = mfma4(f4e(q)[u], f4e(x[p])[u], acc[p]);
      }
    }
    // ---- softmax backward per head: dl = a (dA + dze.e - sum a (dA + dze.e)); tiles: P (shared columns) and dL ----
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const float4 dze = *reinterpret_cast<const float4*>(DZ + (il * 4 + h) * ABD_DZL + F);
      float dot = 0.0f;
#pragma unroll
      for (int p = 0; p < NPR; ++p) {
        const float w = av[p][h];
        const float4 e = efv[p];
        // masked slots (a == 0) may carry 5e5 / NaN edge features: never multiply them
        const float dA = (w != 0.0f) ? acc[p][h] + fmaf(dze.x, e.x, fmaf(dze.y, e.y, fmaf(dze.z, e.z, dze.w * e.w))) : 0.0f;
        acc[p][h] = dA;
        dot = fmaf(w, dA, dot);
      }
      dot = grp8_sum(dot);
#pragma unroll
      for (int p = 0; p < NPR; ++p) {
        acc[p][h] = av[p][h] * (acc[p][h] - dot);
        DL[(il * 4 + h) * PL + p * 8 + sub] = acc[p][h];
        if (p < PS) PT[(il * 4 + h) * PTL + p * 8 + sub] = av[p][h];
      }
    }
    // ---- dQt = dL Xs ----
    {
      float pa[PC];
#pragma unroll
      for (int k = 0; k < PC / 4; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(DL + (il * 4 + hA) * PL + k * 4);
        pa[k * 4] = v.x; pa[k * 4 + 1] = v.y; pa[k * 4 + 2] = v.z; pa[k * 4 + 3] = v.w;
      }
      f32x4g az[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) az[q] = f32x4g{0.f, 0.f, 0.f, 0.f};
      const int fq = (g2 * 4 + c) * 4;
#pragma unroll
      for (int kk = 0; kk < NSP; ++kk) {
        const float4 b = *reinterpret_cast<const float4*>(XS + kk * ABD_XL + fq);
#pragma unroll
        for (int q = 0; q < 4; ++q) az[q] = mfma4(pa[kk], f4e(b)[q], az[q]);
      }
      if constexpr (HITS) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          const float4 b = *reinterpret_cast<const float4*>(XH + (il * 8 + m) * ABD_XL + il * 8 + fq);
#pragma unroll
          for (int q = 0; q < 4; ++q) az[q] = mfma4(pa[NSP + m], f4e(b)[q], az[q]);
        }
      }
      if (live) {
        float* dq = a.dqt + ((size_t)g * n + i) * H * F;
#pragma unroll
        for (int h = 0; h < 4; ++h)
          if (h < H) *reinterpret_cast<float4*>(dq + h * F + fq) = make_float4(az[0][h], az[1][h], az[2][h], az[3][h]);
      }
    }
    if (want_dx) {
      // ---- shared nodes: dXs += P^T dZx + dL^T Qt over the 32 (agent, head) rows of the batch ----
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const int prow = ks * 4 + lq;                          // = agent ks of the batch, head lq
        float bD[2];
#pragma unroll
        for (int ft = 0; ft < 2; ++ft) bD[ft] = DZ[prow * ABD_DZL + ft * 16 + li];
#pragma unroll
        for (int rt = 0; rt < RTS; ++rt) {
          const int col = (rt * 16 + li < NSP) ? rt * 16 + li : NSP - 1;      // tile rows past NSP are never stored
          const float aP = PT[prow * PTL + col], aL = DL[prow * PL + col];
#pragma unroll
          for (int ft = 0; ft < 2; ++ft) {
            dxs[rt][ft] = __builtin_amdgcn_mfma_f32_16x16x4f32(aP, bD[ft], dxs[rt][ft], 0, 0, 0);
            dxs[rt][ft] = __builtin_amdgcn_mfma_f32_16x16x4f32(aL, bQ[ks][ft], dxs[rt][ft], 0, 0, 0);
          }
        }
      }
      // ---- hit nodes: block (il, g2) = 4 hits x 4 feature octets, k = (head, P | dL) ----
      if constexpr (HITS) {
        if (a.dXo != nullptr || (XOF && a.dwo_slab != nullptr)) {
          f32x4g af[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) af[q] = f32x4g{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int h = 0; h < 4; ++h) {
            const float4 d0 = *reinterpret_cast<const float4*>(DZ + (il * 4 + h) * ABD_DZL + c * 8);
            const float4 d1 = *reinterpret_cast<const float4*>(DZ + (il * 4 + h) * ABD_DZL + c * 8 + 4);
            const float pw = av[PS][h], dw = acc[PS][h];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              af[q] = mfma4(pw, f4e(d0)[q], af[q]);
              af[q + 4] = mfma4(pw, f4e(d1)[q], af[q + 4]);
              af[q] = mfma4(dw, f4e(q8[h][0])[q], af[q]);
              af[q + 4] = mfma4(dw, f4e(q8[h][1])[q], af[q + 4]);
            }
          }
          float4 x0[4], x1[4];
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            const float* xr = XH + (il * 8 + g2 * 4 + m) * ABD_XL + il * 8 + c * 8;
            x0[m] = *reinterpret_cast<const float4*>(xr);
            x1[m] = *reinterpret_cast<const float4*>(xr + 4);
          }
          const bool to_slab = XOF && a.dwo_slab != nullptr;
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            const int hb = ab * 64 + il * 8 + g2 * 4 + m;
            float4 v0 = make_float4(af[0][m], af[1][m], af[2][m], af[3][m]);
            float4 v1 = make_float4(af[4][m], af[5][m], af[6][m], af[7][m]);
            if (a.relu_xo || to_slab) {
              v0.x = x0[m].x > 0.f ? v0.x : 0.f; v0.y = x0[m].y > 0.f ? v0.y : 0.f; v0.z = x0[m].z > 0.f ? v0.z : 0.f; v0.w = x0[m].w > 0.f ? v0.w : 0.f;
              v1.x = x1[m].x > 0.f ? v1.x : 0.f; v1.y = x1[m].y > 0.f ? v1.y : 0.f; v1.z = x1[m].z > 0.f ? v1.z : 0.f; v1.w = x1[m].w > 0.f ? v1.w : 0.f;
            }
            if (to_slab) {            // the hit image becomes the image of its gradient (rows of agents past n: zeros, P and dL are 0 there)
              float* xr = XH + (il * 8 + g2 * 4 + m) * ABD_XL + il * 8 + c * 8;
              *reinterpret_cast<float4*>(xr) = v0;
              *reinterpret_cast<float4*>(xr + 4) = v1;
            } else if (hb < n * 8) {
              float* o = a.dXo + ((size_t)g * (Ns - n) + t.ng + hb) * F + c * 8;
              *reinterpret_cast<float4*>(o) = v0;
              *reinterpret_cast<float4*>(o + 4) = v1;
            }
          }
          if constexpr (XOF) {
            if (to_slab) {
              // dWo^T [f][k] += sum over the batch's 64 hit rows dX[row][f] raw[row][k]  (column 8 of B = 1: the bias gradient)
#pragma unroll
              for (int s_ = 0; s_ < 16; ++s_) {
                const int hr = s_ * 4 + lq;
                const float bv = li < 8 ? rbh[s_] : (li == 8 ? 1.0f : 0.0f);
#pragma unroll
                for (int ft = 0; ft < 2; ++ft) {
                  const float av_ = XH[hr * ABD_XL + (hr >> 3) * 8 + ft * 16 + li];
                  dwacc[ft] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_, bv, dwacc[ft], 0, 0, 0);
                }
              }
            }
          }
        }
      }
    }
  }
  if (want_dx) {
    float xm[RTS][2][4];
#pragma unroll
    for (int rt = 0; rt < RTS; ++rt)
#pragma unroll
      for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int node = rt * 16 + lq * 4 + r;
          xm[rt][ft][r] = (node < NSP) ? XS[node * ABD_XL + ft * 16 + li] : 0.0f;
        }
#pragma unroll
    for (int rt = 0; rt < RTS; ++rt)
#pragma unroll
      for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int node = rt * 16 + lq * 4 + r, f = ft * 16 + li;
          float v = dxs[rt][ft][r];
          if (node < n) a.dXa[((size_t)g * n + node) * F + f] = v + dir[rt][ft][r];
          else if (XOF && a.dwo_slab != nullptr) {
            if (node < NSP) XS[node * ABD_XL + f] = (node < NSH && xm[rt][ft][r] > 0.0f) ? v : 0.0f;   // gradient image of the shared other rows
          } else if (node < NSH && a.dXo != nullptr) {
            if (a.relu_xo) v = (xm[rt][ft][r] > 0.0f) ? v : 0.0f;
            a.dXo[((size_t)g * (Ns - n) + (node - n)) * F + f] = v;
          }
        }
    if constexpr (XOF) {
      if (a.dwo_slab != nullptr) {
        const int nsho = NSH - n;
#pragma unroll
        for (int s_ = 0; s_ < NSP / 4; ++s_) {
          const int o = s_ * 4 + lq;
          const float bv = li < 8 ? rbg[s_] : (li == 8 ? 1.0f : 0.0f);
#pragma unroll
          for (int ft = 0; ft < 2; ++ft) {
            const int node = n + o;
            const float av_ = (o < nsho && node < NSP) ? XS[(node < NSP ? node : NSP - 1) * ABD_XL + ft * 16 + li] : 0.0f;
            dwacc[ft] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_, bv, dwacc[ft], 0, 0, 0);
          }
        }
        if (li <= 8) {
          float* o = a.dwo_slab + (size_t)g * ABD_DW_STRIDE + li * 32 + lq * 4;
#pragma unroll
          for (int ft = 0; ft < 2; ++ft)
            *reinterpret_cast<float4*>(o + ft * 16) = make_float4(dwacc[ft][0], dwacc[ft][1], dwacc[ft][2], dwacc[ft][3]);
        }
      }
    }
  }
}

// the block-diagonal kernels apply when the private nodes are exactly 8 LiDAR hits per agent (or there are none)
static bool attn_bd_shape(const Topo& t, int F, int H, int& PS, bool& hits) {
  if (F != 32 || H < 1 || H > 4 || getenv("DGPPO_ATTN_NO_BD")) return false;
  const int n_priv = t.Ns - t.n - t.ng;
  hits = t.lidar && n_priv > 0;
  if (hits && (t.per != 8 || n_priv != t.n * 8)) return false;
  const int nsh = hits ? t.n + t.ng : t.Ns;
  PS = (nsh + 7) / 8;
  return PS >= 1 && PS <= 4;
}
template <int PS, bool HITS, int AB, bool XOF>
static void launch_attn_bd_x(const AttnArgs& a, hipStream_t s, bool bwd) {
  using L = AbdLds<PS, HITS>;
  if (bwd) { hipLaunchKernelGGL((attn_bwd_bd_kernel<PS, HITS, AB, XOF>), dim3(a.G), dim3(64), sizeof(float) * L::BWD, s, a); return; }
  if constexpr (AB == 1) {
    if (!getenv("DGPPO_ATTN_NO_PERSIST")) {
      // persistent waves with the next graph's loads in flight: as many 2-wave workgroups as fit the chip (LDS-limited), unless
      // the launch is small enough that every graph gets its own wave anyway
      static thread_local int cap = 0;
      if (cap == 0) {
        int per_cu = 0, dev = 0, cus = 256;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(&attn_fwd_bdp_kernel<PS, HITS, XOF>), 128,
                                                         sizeof(float) * 2 * L::FWD) != hipSuccess || per_cu < 1) per_cu = 1;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            cus < 1) cus = 256;
        cap = per_cu * cus;
      }
      const int wgs = (a.G + 1) / 2;
      if (const char* force = getenv("DGPPO_ATTN_PERSIST_WGS")) {     // tests: walk several graphs per wave on a tiny launch
        const int fw = atoi(force);
        if (fw > 0) {
          hipLaunchKernelGGL((attn_fwd_bdp_kernel<PS, HITS, XOF>), dim3(fw < wgs ? fw : wgs), dim3(128), sizeof(float) * 2 * L::FWD, s, a);
          return;
        }
      }
      if (wgs > 2 * cap) {        // (smaller launches: one graph per wave in the leaner kernel below — more waves per CU)
        hipLaunchKernelGGL((attn_fwd_bdp_kernel<PS, HITS, XOF>), dim3(cap), dim3(128), sizeof(float) * 2 * L::FWD, s, a);
        return;
      }
    }
  }
  hipLaunchKernelGGL((attn_fwd_bd_kernel<PS, HITS, AB, XOF>), dim3((a.G + 1) / 2), dim3(128), sizeof(float) * 2 * L::FWD, s, a);
}
template <int PS, bool HITS, int AB>
static void launch_attn_bd_one(const AttnArgs& a, hipStream_t s, bool bwd) {
  if (a.Xo_raw != nullptr) launch_attn_bd_x<PS, HITS, AB, true>(a, s, bwd);
  else launch_attn_bd_x<PS, HITS, AB, false>(a, s, bwd);
}
static bool launch_attn_bd(const AttnArgs& a, hipStream_t s, bool bwd) {
  int PS = 0;
  bool hits = false;
  if (!attn_bd_shape(a.t, a.F, a.H, PS, hits)) return false;
  const int AB = (a.t.n + 7) / 8;                   // agent batches; the shared nodes include the agents, so AB <= (PS + 1) / 2 <= 2
  switch ((PS * 2 + (hits ? 1 : 0)) * 4 + AB) {
#define DGPPO_BD(P, HT, B) case ((P) * 2 + (HT)) * 4 + (B): launch_attn_bd_one<P, (HT) != 0, B>(a, s, bwd); return true;
    DGPPO_BD(1, 0, 1) DGPPO_BD(1, 1, 1) DGPPO_BD(2, 0, 1) DGPPO_BD(2, 1, 1)
    DGPPO_BD(3, 0, 1) DGPPO_BD(3, 1, 1) DGPPO_BD(3, 0, 2) DGPPO_BD(3, 1, 2)
    DGPPO_BD(4, 0, 1) DGPPO_BD(4, 1, 1) DGPPO_BD(4, 0, 2) DGPPO_BD(4, 1, 2)
#undef DGPPO_BD
    default: return false;
  }
}

// dispatch over the compile-time tile counts; returns false if the shape has no instantiation
template <int F, int CT, int NP>
static bool launch_attn_wave_sj(const AttnArgs& a, int SJ, int grid, hipStream_t s, bool bwd) {
  const size_t smem = sizeof(float) * 4 * ((NP + 1) / 2) * 16 * (CT * 16 + 1);
#define DGPPO_SJ(J)                                                                                              \
  case J:                                                                                                        \
    if (bwd) hipLaunchKernelGGL((attn_bwd_wave_kernel<F, CT, NP, J>), dim3(grid), dim3(256), smem, s, a);        \
    else hipLaunchKernelGGL((attn_fwd_wave_kernel<F, CT, NP, J>), dim3(grid), dim3(256), smem, s, a);            \
    return true;
  switch (SJ) {
    DGPPO_SJ(1) DGPPO_SJ(2) DGPPO_SJ(3) DGPPO_SJ(4)
    default: return false;
  }
#undef DGPPO_SJ
}
template <int F, int CT>
static bool launch_attn_wave_np(const AttnArgs& a, int NP, int SJ, int grid, hipStream_t s, bool bwd) {
  switch (NP) {
    case 1: return launch_attn_wave_sj<F, CT, 1>(a, SJ, grid, s, bwd);
    case 2: return launch_attn_wave_sj<F, CT, 2>(a, SJ, grid, s, bwd);
    case 3: return launch_attn_wave_sj<F, CT, 3>(a, SJ, grid, s, bwd);
    case 4: return launch_attn_wave_sj<F, CT, 4>(a, SJ, grid, s, bwd);
    default: return false;
  }
}
template <int F>
static bool launch_attn_wave(const AttnArgs& a, int CT, int NP, int SJ, int grid, hipStream_t s, bool bwd = false) {
  constexpr int FQ = F / 4, FT = (F + 15) / 16;
  constexpr int CTMAX = (ATW_BX / FQ) < (ATW_BZ / (4 * FT)) ? (ATW_BX / FQ) : (ATW_BZ / (4 * FT));
  switch (CT) {
#define DGPPO_CASE(C) case C: if constexpr (C <= CTMAX) return launch_attn_wave_np<F, C>(a, NP, SJ, grid, s, bwd); else return false;
    DGPPO_CASE(1) DGPPO_CASE(2) DGPPO_CASE(3) DGPPO_CASE(4) DGPPO_CASE(5) DGPPO_CASE(6)
#undef DGPPO_CASE
    default: return false;
  }
}

__global__ void __launch_bounds__(256) attn_bwd_kernel(AttnArgs a) {
  extern __shared__ float sm[];
  const Topo& t = a.t;
  const AttnDims d = attn_dims(t, a.F, a.H);
  const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = d.n, S = d.S, Ns = d.Ns, F = d.F, H = d.H, Fl = d.Fl, Ll = d.Ll, nH = d.nH;
  const int Wd = F + 4, Zl = F + 5, Sp = (S + 3) & ~3;
  float* s_x = sm;
  float* s_q = s_x + d.CT * 16 * Fl;
  float* s_P = s_q + d.RT * 16 * Fl;
  float* s_e = sm + (((d.CT * 16 * Fl + d.RT * 16 * Fl + d.RT * 16 * Ll) + 3) & ~3);   // float4-aligned
  float* s_c = s_e + n * S * 4;           // compact [nH][Sp]: dA at the slots -> dl     (the forward's s_m + s_a space)
  float* s_a = s_c + nH * Sp;             // compact [nH][Sp]: attention weights
  float* s_D = s_a + nH * Sp;
  float* s_z = s_D + d.RT * 16 * Ll;
  const float* dzc = a.dzcat + (size_t)g * n * a.Kp;
  {
    const int F4 = F >> 2, W4 = Wd >> 2;       // Wd = F + 4, Kp and F are multiples of 4 -> 16-byte aligned pieces
    const float4* xa = reinterpret_cast<const float4*>(a.Xa + (size_t)g * n * F);
    const float4* xo = reinterpret_cast<const float4*>(a.Xo + (size_t)g * (Ns - n) * F);
    const float4* q4 = reinterpret_cast<const float4*>(a.qt + (size_t)g * nH * F);
    const float4* e4 = reinterpret_cast<const float4*>(a.efeat + (size_t)g * n * S * 4);
    const float* at = a.attn + (size_t)g * n * S * H;
    const int nA = n * F4, nX = Ns * F4, nXp = d.CT * 16 * F4, nQ = nH * F4, nQp = d.RT * 16 * F4, nE = n * S;
    const int nZp = d.RT * 16 * W4, nAt = n * S * H;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    auto ldz = [&](int idx) {
      const int row = idx / W4, q = idx - row * W4;
      if (row >= nH) return z4;
      const int i = row / H, h = row - i * H;
      return *reinterpret_cast<const float4*>(dzc + i * a.Kp + F + h * Wd + 4 * q);
    };
    float4 vx[4], vq[2], ve[2], vz[2];
    float va[3];
#pragma unroll
    for (int it = 0; it < 4; ++it) { const int idx = tid + it * 256; vx[it] = (idx < nA) ? xa[idx] : ((idx < nX) ? xo[idx - nA] : z4); }
#pragma unroll
    for (int it = 0; it < 2; ++it) { const int idx = tid + it * 256; vq[it] = (idx < nQ) ? q4[idx] : z4; ve[it] = (idx < nE) ? e4[idx] : z4; vz[it] = (idx < nZp) ? ldz(idx) : z4; }
#pragma unroll
    for (int it = 0; it < 3; ++it) { const int idx = tid + it * 256; va[it] = (idx < nAt) ? at[idx] : 0.0f; }
#pragma unroll
    for (int it = 0; it < 4; ++it) { const int idx = tid + it * 256; if (idx < nXp) { const int nd = idx / F4, q = idx - nd * F4; put4(s_x + nd * Fl + 4 * q, vx[it]); } }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int idx = tid + it * 256;
      if (idx < nQp) { const int row = idx / F4, q = idx - row * F4; put4(s_q + row * Fl + 4 * q, vq[it]); }
      if (idx < nE) reinterpret_cast<float4*>(s_e)[idx] = ve[it];
      if (idx < nZp) { const int row = idx / W4, q = idx - row * W4; put4(s_z + row * Zl + 4 * q, vz[it]); }
    }
    auto puta = [&](int idx, float v) { const int h = idx % H, is = idx / H, i = is / S, s = is - i * S; s_a[(i * H + h) * Sp + s] = v; };
#pragma unroll
    for (int it = 0; it < 3; ++it) { const int idx = tid + it * 256; if (idx < nAt) puta(idx, va[it]); }
    for (int idx = tid + 1024; idx < nXp; idx += 256) { const int nd = idx / F4, q = idx - nd * F4; put4(s_x + nd * Fl + 4 * q, (idx < nA) ? xa[idx] : ((idx < nX) ? xo[idx - nA] : z4)); }
    for (int idx = tid + 512; idx < nQp; idx += 256) { const int row = idx / F4, q = idx - row * F4; put4(s_q + row * Fl + 4 * q, (idx < nQ) ? q4[idx] : z4); }
    for (int idx = tid + 512; idx < nE; idx += 256) reinterpret_cast<float4*>(s_e)[idx] = e4[idx];
    for (int idx = tid + 512; idx < nZp; idx += 256) { const int row = idx / W4, q = idx - row * W4; put4(s_z + row * Zl + 4 * q, ldz(idx)); }
    for (int idx = tid + 768; idx < nAt; idx += 256) puta(idx, at[idx]);
    for (int idx = tid; idx < d.RT * 16 * Ll; idx += 256) s_P[idx] = 0.0f;
    if (Sp != S) for (int idx = tid; idx < nH; idx += 256) for (int s = S; s < Sp; ++s) s_a[idx * Sp + s] = 0.0f;
  }
  __syncthreads();
  // P scatter (for dXs) and dA = dZx * Xs^T
  for (int idx = tid; idx < nH * S; idx += 256) {
    const int pair = idx / S, s = idx - pair * S, i = pair / H;
    const float av = s_a[pair * Sp + s];
    if (av != 0.0f) s_P[pair * Ll + sender_node(t, i, s)] = av;
  }
  mfma_tiles(d.RT, d.CT, F / 4, wave, lane,
             [&](int row, int k) { return s_z[row * Zl + k]; },
             [&](int k, int col) { return s_x[col * Fl + k]; },
             [&](int row, int col, float v) { s_D[row * Ll + col] = v; });
  __syncthreads();
  // gather dA at the slots (+ the edge-feature term) into the compact rows
  for (int idx = tid; idx < nH * Sp; idx += 256) {
    const int pair = idx / Sp, s = idx - pair * Sp, i = pair / H;
    float dA = 0.0f;
    if (s < S && s_a[idx] != 0.0f) {
      dA = s_D[pair * Ll + sender_node(t, i, s)];
      const float* dz = s_z + pair * Zl + F;
      const float4 e = reinterpret_cast<const float4*>(s_e)[i * S + s];
      dA = fmaf(dz[0], e.x, fmaf(dz[1], e.y, fmaf(dz[2], e.z, fmaf(dz[3], e.w, dA))));
    }
    s_c[idx] = dA;
  }
  __syncthreads();
  // 8 lanes per (agent, head): softmax backward dl = a * (dA - sum_s a dA); then every lane helps clearing s_D
  for (int p0 = 0; p0 < nH; p0 += 32) {
    const int pair = p0 + (tid >> 3), sub = tid & 7;
    const bool live = pair < nH;
    float av[ATT_SMAX / 8], dA[ATT_SMAX / 8];
    float dot = 0.0f;
#pragma unroll
    for (int j = 0; j < ATT_SMAX / 8; ++j) {
      const int sl = sub + 8 * j;
      av[j] = (live && sl < S) ? s_a[pair * Sp + sl] : 0.0f;
      dA[j] = (live && sl < S) ? s_c[pair * Sp + sl] : 0.0f;
      dot = fmaf(av[j], dA[j], dot);
    }
    dot = grp8_sum(dot);
#pragma unroll
    for (int j = 0; j < ATT_SMAX / 8; ++j) {
      const int sl = sub + 8 * j;
      if (live && sl < S) s_c[pair * Sp + sl] = av[j] * (dA[j] - dot);
    }
  }
  for (int idx = tid; idx < d.RT * 16 * Ll; idx += 256) s_D[idx] = 0.0f;
  __syncthreads();
  for (int idx = tid; idx < nH * S; idx += 256) {
    const int pair = idx / S, s = idx - pair * S, i = pair / H;
    const float dl = s_c[pair * Sp + s];
    if (dl != 0.0f) s_D[pair * Ll + sender_node(t, i, s)] = dl;
  }
  __syncthreads();
  // dQt = dL * Xs
  float* dq = a.dqt + (size_t)g * nH * F;
  mfma_tiles(d.RT, (F + 15) / 16, d.CT * 4, wave, lane,
             [&](int row, int k) { return s_D[row * Ll + k]; },
             [&](int k, int col) { return (col < F) ? s_x[k * Fl + col] : 0.0f; },
             [&](int row, int col, float v) { if (row < nH && col < F) dq[row * F + col] = v; });
  // dXs = P^T * dZx + dL^T * Qt  (+ direct x_i part for agents)
  if (a.dXa != nullptr) {
    const int li = lane & 15, lq = lane >> 4;
    const int FT = (F + 15) / 16;
    for (int tile = wave; tile < d.CT * FT; tile += 4) {
      const int rt = tile / FT, ct = tile - rt * FT;
      f32x4g acc = {0.f, 0.f, 0.f, 0.f};
      const int nd_a = rt * 16 + li, col_b = ct * 16 + li;
#pragma unroll 4
      for (int k4 = 0; k4 < d.RT * 4; ++k4) {
        const int kk = k4 * 4 + lq;   // (agent, head) row
        const float bz = (col_b < F) ? s_z[kk * Zl + col_b] : 0.0f;
        const float bq = (col_b < F) ? s_q[kk * Fl + col_b] : 0.0f;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(s_P[kk * Ll + nd_a], bz, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(s_D[kk * Ll + nd_a], bq, acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int nd = rt * 16 + lq * 4 + r, f = ct * 16 + li;
        if (nd >= Ns || f >= F) continue;
        if (nd < n) a.dXa[((size_t)g * n + nd) * F + f] = acc[r] + dzc[nd * a.Kp + f];
        else if (a.dXo != nullptr) a.dXo[((size_t)g * (Ns - n) + (nd - n)) * F + f] = acc[r];
      }
    }
  }
}

static int32_t attn_check(const dgppo_env_cfg* cfg, int F, int H, int Kp, int G, AttnArgs& a) {
  int32_t rc = dgppo_validate_cfg(cfg);
  if (rc) return rc;
  a.t = make_topo(*cfg);
  DGPPO_REQUIRE(F >= 1 && F <= 64 && H >= 1 && H <= 8, "attn: bad F=%d H=%d", F, H);
  DGPPO_REQUIRE(Kp >= F + H * (F + 4) + 1, "attn: Kp=%d too small", Kp);
  DGPPO_REQUIRE(G >= 0, "attn: G < 0");
  a.F = F; a.H = H; a.Kp = Kp; a.G = G;
  return 0;
}

extern "C" int32_t dgppo_attn_fwd(const dgppo_env_cfg* cfg, int32_t F, int32_t H, int32_t Kp, const float* qt,
                                  const float* Xa, const float* Xo, const float* efeat, const float* emask, float* zcat,
                                  float* attn, int32_t G, void* stream) {
  AttnArgs a{};
  int32_t rc = attn_check(cfg, F, H, Kp, G, a);
  if (rc) return rc;
  if (G == 0) return 0;
  DGPPO_REQUIRE(qt && Xa && efeat && emask && zcat, "attn_fwd: NULL operand");      // attn == NULL: inference, weights not kept
  DGPPO_REQUIRE(a.t.Ns == a.t.n || Xo, "attn_fwd: Xo is NULL");
  a.qt = qt; a.Xa = Xa; a.Xo = Xo; a.efeat = efeat; a.emask = emask; a.zcat = zcat; a.attn = attn;
  const Topo& t = a.t;
  const size_t smem = sizeof(float) * ((size_t)t.Ns * (F + 1) + (size_t)t.n * H * (F + 1) + (size_t)t.n * t.S * 5 +
                                       (size_t)t.n * t.S * H);
  DGPPO_REQUIRE(smem <= 64 * 1024, "attn_fwd: graph too large for LDS (%zu B)", smem);
  const AttnDims d = attn_dims(t, F, H);
  const size_t msmem = attn_mfma_smem(d, false);
  const bool mfma_ok = (F & 3) == 0 && t.S <= 64 && msmem <= 64 * 1024 && !getenv("DGPPO_ATTN_VALU");
  // one wave per graph when the shape has an instantiation (fragments within the register budget)
  bool launched = false;
  if (mfma_ok && (Kp & 3) == 0 && t.n * H * H < 65536 && !getenv("DGPPO_ATTN_BLOCK")) {
    const int grid = (G + 3) / 4;
    // instantiated for the widths the reference's defaults produce (node features padded to 8, msg_dim 32); other
    // multiples of 4 take the workgroup-per-graph kernels below
    // narrow first layer: slot-sparse VALU kernel (one memory round trip per graph); DGPPO_ATTN_DENSE8 forces the MFMA form
    if (F == 8 && !getenv("DGPPO_ATTN_DENSE8")) launched = launch_attn_slot8(a, grid, (hipStream_t)stream, false);
    if (!launched && F == 32) launched = launch_attn_bd(a, (hipStream_t)stream, false);
    if (launched) {}
    else if (F == 8) launched = launch_attn_wave<8>(a, d.CT, (d.nH + 7) / 8, (t.S + 7) / 8, grid, (hipStream_t)stream);
    else if (F == 32) launched = launch_attn_wave<32>(a, d.CT, (d.nH + 7) / 8, (t.S + 7) / 8, grid, (hipStream_t)stream);
  }
  if (!launched) {       // workgroup-per-graph fallbacks: MFMA for F % 4 == 0, plain VALU otherwise
    if (mfma_ok) hipLaunchKernelGGL(attn_fwd_kernel, dim3(G), dim3(256), msmem, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(attn_fwd_valu_kernel, dim3(G), dim3(256), smem, (hipStream_t)stream, a);
  }
  DGPPO_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t dgppo_attn_bwd(const dgppo_env_cfg* cfg, int32_t F, int32_t H, int32_t Kp, const float* dzcat,
                                  const float* attn, const float* qt, const float* Xa, const float* Xo,
                                  const float* efeat, float* dqt, float* dXa, float* dXo, int32_t relu_xo, int32_t G,
                                  void* stream) {
  AttnArgs a{};
  int32_t rc = attn_check(cfg, F, H, Kp, G, a);
  if (rc) return rc;
  if (G == 0) return 0;
  DGPPO_REQUIRE(dzcat && attn && qt && Xa && efeat && dqt, "attn_bwd: NULL operand");
  a.relu_xo = 0;
  DGPPO_REQUIRE(a.t.Ns == a.t.n || Xo, "attn_bwd: Xo is NULL");
  DGPPO_REQUIRE(!(dXo && !dXa), "attn_bwd: dXo needs dXa");
  a.dzcat = dzcat; a.attn = (float*)attn; a.qt = qt; a.Xa = Xa; a.Xo = Xo; a.efeat = efeat; a.dqt = dqt; a.dXa = dXa; a.dXo = dXo;
  const Topo& t = a.t;
  const size_t smem = sizeof(float) * ((size_t)t.Ns * (F + 1) + (size_t)t.n * H * (F + 1) + (size_t)t.n * t.S * 4 +
                                       2 * (size_t)t.n * t.S * H + (size_t)t.n * H * (F + 5));
  DGPPO_REQUIRE(smem <= 64 * 1024, "attn_bwd: graph too large for LDS (%zu B)", smem);
  const AttnDims d = attn_dims(t, F, H);
  const size_t msmem = attn_mfma_smem(d, true);
  // one wave per graph when the shape has an instantiation (same conditions as the forward)
  bool launched = false;
  if ((F & 3) == 0 && t.S <= 64 && (Kp & 3) == 0 && t.n * H * H < 65536 && !getenv("DGPPO_ATTN_VALU") &&
      !getenv("DGPPO_ATTN_BLOCK")) {
    const int grid = (G + 3) / 4, NP = (d.nH + 7) / 8, SJ = (t.S + 7) / 8;
    a.relu_xo = (relu_xo && dXo) ? 1 : 0;          // fused into the wave kernel's dXo store
    if (F == 8 && !dXa && !getenv("DGPPO_ATTN_DENSE8")) launched = launch_attn_slot8(a, grid, (hipStream_t)stream, true);
    if (!launched && F == 32) launched = launch_attn_bd(a, (hipStream_t)stream, true);
    if (launched) {}
    else if (F == 8) launched = launch_attn_wave<8>(a, d.CT, NP, SJ, grid, (hipStream_t)stream, true);
    else if (F == 32) launched = launch_attn_wave<32>(a, d.CT, NP, SJ, grid, (hipStream_t)stream, true);
    a.relu_xo = 0;
  }
  if (!launched) {       // workgroup-per-graph fallbacks; among them the VALU kernel wins for narrow layers (measured)
    if ((F & 3) == 0 && F >= 16 && t.S <= 64 && msmem <= 64 * 1024 && !getenv("DGPPO_ATTN_VALU"))
      hipLaunchKernelGGL(attn_bwd_kernel, dim3(G), dim3(256), msmem, (hipStream_t)stream, a);
    else
      hipLaunchKernelGGL(attn_bwd_valu_kernel, dim3(G), dim3(256), smem, (hipStream_t)stream, a);
    DGPPO_LAUNCH_CHECK();
    if (relu_xo && dXo)    // the fallback kernels do not fuse the ReLU mask of the other nodes' gradient: separate pass
      return dgppo_relu_bwd(dXo, Xo, (int64_t)G * (t.Ns - t.n) * F, stream);
    return 0;
  }
  DGPPO_LAUNCH_CHECK();
  return 0;
}

// ---- the same layer with the other nodes' rows recomputed inside the kernel (block-diagonal kernels only) ------------------
// Xo = relu(Xo_raw Wo + bo), Xo_raw [G*(Ns-n), 8] = the padded raw features of goals / hits / obstacles, Wo [8, ldwo >= 32] =
// the first 8 rows of the previous layer's update weight, bo [32] its bias: what the reference computes for nodes without
// incoming edges (gnn.py:109-111 with aggr = 0) and then feeds to the next GraphTransformer layer as sender rows
// (gnn.py:85-117).  dgppo_attn_xo_supported tells the caller whether the topology has such a kernel; if not, materialise Xo
// (dgppo_dense_fwd) and call dgppo_attn_fwd / dgppo_attn_bwd.
static bool attn_xo_ok(const Topo& t, int F, int H, int Kp) {
  int PS = 0;
  bool hits = false;
  return t.Ns > t.n && (Kp & 3) == 0 && t.S <= 64 && t.n * H * H < 65536 && !getenv("DGPPO_ATTN_VALU") && !getenv("DGPPO_ATTN_BLOCK") &&
         !getenv("DGPPO_ATTN_NO_XO") && attn_bd_shape(t, F, H, PS, hits);
}
extern "C" int32_t dgppo_attn_xo_supported(const dgppo_env_cfg* cfg, int32_t F, int32_t H, int32_t Kp) {
  AttnArgs a{};
  if (attn_check(cfg, F, H, Kp, 0, a)) return 0;
  return attn_xo_ok(a.t, F, H, Kp) ? 1 : 0;
}
extern "C" int32_t dgppo_attn_fwd_xo(const dgppo_env_cfg* cfg, int32_t F, int32_t H, int32_t Kp, const float* qt, const float* Xa,
                                     const float* Xo_raw, const float* Wo, int32_t ldwo, const float* bo, const float* efeat,
                                     const float* emask, float* zcat, float* attn, int32_t G, void* stream) {
  AttnArgs a{};
  int32_t rc = attn_check(cfg, F, H, Kp, G, a);
  if (rc) return rc;
  if (G == 0) return 0;
  DGPPO_REQUIRE(qt && Xa && Xo_raw && Wo && bo && efeat && emask && zcat, "attn_fwd_xo: NULL operand");
  DGPPO_REQUIRE(ldwo >= 32, "attn_fwd_xo: ldwo=%d < 32", ldwo);
  DGPPO_REQUIRE(attn_xo_ok(a.t, F, H, Kp), "attn_fwd_xo: no fused kernel for this topology (ask dgppo_attn_xo_supported)");
  a.qt = qt; a.Xa = Xa; a.Xo = nullptr; a.efeat = efeat; a.emask = emask; a.zcat = zcat; a.attn = attn;
  a.Xo_raw = Xo_raw; a.Wo = Wo; a.ldwo = ldwo; a.bo = bo;
  const bool launched = launch_attn_bd(a, (hipStream_t)stream, false);
  DGPPO_REQUIRE(launched, "attn_fwd_xo: dispatch failed");
  DGPPO_LAUNCH_CHECK();
  return 0;
}
extern "C" int32_t dgppo_attn_bwd_xo(const dgppo_env_cfg* cfg, int32_t F, int32_t H, int32_t Kp, const float* dzcat,
                                     const float* attn, const float* qt, const float* Xa, const float* Xo_raw, const float* Wo,
                                     int32_t ldwo, const float* bo, const float* efeat, float* dqt, float* dXa, float* dXo,
                                     int32_t relu_xo, int32_t G, void* stream) {
  AttnArgs a{};
  int32_t rc = attn_check(cfg, F, H, Kp, G, a);
  if (rc) return rc;
  if (G == 0) return 0;
  DGPPO_REQUIRE(dzcat && attn && qt && Xa && Xo_raw && Wo && bo && efeat && dqt, "attn_bwd_xo: NULL operand");
  DGPPO_REQUIRE(ldwo >= 32, "attn_bwd_xo: ldwo=%d < 32", ldwo);
  DGPPO_REQUIRE(!(dXo && !dXa), "attn_bwd_xo: dXo needs dXa");
  DGPPO_REQUIRE(attn_xo_ok(a.t, F, H, Kp), "attn_bwd_xo: no fused kernel for this topology (ask dgppo_attn_xo_supported)");
  a.dzcat = dzcat; a.attn = (float*)attn; a.qt = qt; a.Xa = Xa; a.Xo = nullptr; a.efeat = efeat; a.dqt = dqt; a.dXa = dXa; a.dXo = dXo;
  a.relu_xo = (relu_xo && dXo) ? 1 : 0;
  a.Xo_raw = Xo_raw; a.Wo = Wo; a.ldwo = ldwo; a.bo = bo;
  const bool launched = launch_attn_bd(a, (hipStream_t)stream, true);
  DGPPO_REQUIRE(launched, "attn_bwd_xo: dispatch failed");
  DGPPO_LAUNCH_CHECK();
  return 0;
}

// sum of the per-graph slabs of attn_bwd_bd_kernel<.., XOF>: dWo [8, lddwo] += , dbo [32] +=
__global__ void __launch_bounds__(320) attn_xo_dw_reduce_kernel(const float* slab, int G, float* dWo, int lddwo, float* dbo) {
  const int e = threadIdx.x;                      // 0..287: k * 32 + f, then the 32 bias entries
  if (e >= 288) return;
  const int per = (G + gridDim.x - 1) / gridDim.x;
  const int g0 = blockIdx.x * per, g1 = min(G, g0 + per);
  float sacc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int g = g0;
  for (; g + 7 < g1; g += 8) {                     // eight loads in flight per thread
#pragma unroll
    for (int u = 0; u < 8; ++u) sacc[u] += slab[(size_t)(g + u) * ABD_DW_STRIDE + e];
  }
  for (; g < g1; ++g) sacc[0] += slab[(size_t)g * ABD_DW_STRIDE + e];
  const float sum = ((sacc[0] + sacc[1]) + (sacc[2] + sacc[3])) + ((sacc[4] + sacc[5]) + (sacc[6] + sacc[7]));
  if (g0 >= g1) return;
  if (e < 256) atomicAdd(dWo + (size_t)(e >> 5) * lddwo + (e & 31), sum);
  else atomicAdd(dbo + (e - 256), sum);
}
extern "C" int64_t dgppo_attn_xo_workspace_bytes(int32_t G) { return G < 0 ? 0 : (int64_t)G * ABD_DW_STRIDE * (int64_t)sizeof(float); }
// dgppo_attn_bwd_xo that CONSUMES the gradient of the recomputed rows instead of writing it: dWo [8, lddwo] += Xo_raw^T dpre,
// dbo [32] += colsum dpre with dpre = relu'(Xo) * dXo — the weight gradient jax.grad assigns to the previous layer's update Dense
// for the nodes without incoming edges (gnn.py:109-111).  workspace: dgppo_attn_xo_workspace_bytes(G) bytes, caller-owned.
extern "C" int32_t dgppo_attn_bwd_xo_dw(const dgppo_env_cfg* cfg, int32_t F, int32_t H, int32_t Kp, const float* dzcat,
                                        const float* attn, const float* qt, const float* Xa, const float* Xo_raw, const float* Wo,
                                        int32_t ldwo, const float* bo, const float* efeat, float* dqt, float* dXa, float* dWo,
                                        int32_t lddwo, float* dbo, float* workspace, int64_t workspace_bytes, int32_t G,
                                        void* stream) {
  AttnArgs a{};
  int32_t rc = attn_check(cfg, F, H, Kp, G, a);
  if (rc) return rc;
  if (G == 0) return 0;
  DGPPO_REQUIRE(dzcat && attn && qt && Xa && Xo_raw && Wo && bo && efeat && dqt && dXa && dWo && dbo && workspace,
                "attn_bwd_xo_dw: NULL operand");
  DGPPO_REQUIRE(ldwo >= 32 && lddwo >= 32, "attn_bwd_xo_dw: ldwo=%d lddwo=%d < 32", ldwo, lddwo);
  DGPPO_REQUIRE(workspace_bytes >= dgppo_attn_xo_workspace_bytes(G) && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0,
                "attn_bwd_xo_dw: workspace too small or not 16-byte aligned");
  DGPPO_REQUIRE(attn_xo_ok(a.t, F, H, Kp), "attn_bwd_xo_dw: no fused kernel for this topology (ask dgppo_attn_xo_supported)");
  a.dzcat = dzcat; a.attn = (float*)attn; a.qt = qt; a.Xa = Xa; a.Xo = nullptr; a.efeat = efeat; a.dqt = dqt; a.dXa = dXa; a.dXo = nullptr;
  a.relu_xo = 1;
  a.Xo_raw = Xo_raw; a.Wo = Wo; a.ldwo = ldwo; a.bo = bo; a.dwo_slab = workspace;
  const bool launched = launch_attn_bd(a, (hipStream_t)stream, true);
  DGPPO_REQUIRE(launched, "attn_bwd_xo_dw: dispatch failed");
  int blocks = (G + 31) / 32;                      // 32 graphs per workgroup: <= 512 atomics per address at 16 384 graphs
  blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
  hipLaunchKernelGGL(attn_xo_dw_reduce_kernel, dim3(blocks), dim3(320), 0, (hipStream_t)stream, workspace, G, dWo, lddwo, dbo);
  DGPPO_LAUNCH_CHECK();
  return 0;
}
