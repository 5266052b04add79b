// Batched env.step for the LiDAR and MPE families: dynamics -> LiDAR ray-cast -> stable top-k ->
// reward/cost -> padded GraphsTuple, one workgroup per environment.
//
// Reference arithmetic replaced (file:line relative to /root/reference):
//   LidarEnv.step                dgppo/env/lidar_env/base.py:151-174     MPE.step      dgppo/env/mpe/base.py:137-162
//   agent_step_euler             lidar_env/base.py:142-149, lidar_bicycle_target.py:92-111, mpe/base.py:129-135
//   get_lidar / raytracing       dgppo/env/utils.py:49-55,115-136 ; Rectangle.inside/raytracing obstacle.py:62-105
//   get_reward                   lidar_spread.py:35-52, lidar_target.py:35-52, mpe_spread.py:32-49, mpe_target.py:32-49
//   get_cost                     lidar_env/base.py:180-207, mpe/base.py:164-191
//   edge_blocks/get_graph        lidar_spread.py:57-96, lidar_target.py:57-96, mpe_spread.py:51-81, mpe_target.py:51-80,
//                                lidar_env/base.py:227-271, mpe/base.py:211-241 ; to_padded utils/graph.py:35-44,212-247
//
// Built with -ffp-contract=off: every fp32 operation is a single IEEE operation in the same order as
// oracle/env_np.py, so states, rewards, costs, features and therefore all masks/indices are bit-identical.
#include "env_step.h"
#include <stdlib.h>

template <int SD>
__global__ void env_step_kernel(StepArgs a) {
  extern __shared__ float smem[];
  const dgppo_env_cfg& c = a.cfg;
  const int b = blockIdx.x;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int n = c.n_agents, ng = c.n_goals, no = c.n_obs, R = c.n_rays, k = c.top_k;
  const bool lidar = cfg_is_lidar(c), spread = cfg_is_spread(c);
  const int ostride = lidar ? DGPPO_RECT_STRIDE : SD;
  const int n_on = cfg_obs_nodes(c);
  const int kk = lidar ? (no > 0 ? k : 0) : 0;       // hits per agent
  constexpr int ND = SD + 3;

  float* s_seg = smem;                      // no*4*4 per-segment constants (float4-aligned: first in the carve)
  float* s_agent = s_seg + (lidar ? no * 16 : 0);  // n*SD   state at t
  float* s_next = s_agent + n * SD;         // n*SD   state at t+1
  float* s_goal = s_next + n * SD;          // ng*SD
  float* s_act = s_goal + ng * SD;          // n*2    clipped action
  float* s_obst = s_act + n * 2;            // no*ostride
  float* s_hpre = s_obst + no * ostride;    // n*kk*2 hits of graph_t
  float* s_hnext = s_hpre + n * kk * 2;     // n*kk*2 hits of graph_{t+1}
  float* s_alpha = s_hnext + n * kk * 2;    // n*R
  float* s_fa = s_alpha + (lidar ? n * R : 0);  // n*4   state2feat(next agent)
  float* s_fg = s_fa + n * 4;               // ng*4
  const int nrg = cfg_reward_goals(c);      // positions the reward measures against (the goal nodes, or n derived goals)
  float* s_d2g = s_fg + ng * 4;             // nrg
  float* s_an2 = s_d2g + nrg;               // n
  float* s_isin = s_an2 + n;                // n (0/1)
  float* s_md = s_isin + n;                 // n   nearest-neighbour distance of each agent (connectivity cost)
  float* s_rg = s_md + n;                   // nrg*2 reward goal positions
  float* s_ino = s_rg + nrg * 2;            // n*no
  float* s_pair = s_ino + n * no;           // n*n + n*max(kk,no) + nrg*n

  const bool do_dyn = (a.mode == MODE_STEP);
  const bool do_sense = (a.mode != MODE_GRAPH) && lidar && no > 0;

  // ---- phase 0: stage the env's state in LDS -------------------------------------------------
  for (int i = tid; i < n * SD; i += nt) s_agent[i] = a.agent[(size_t)b * n * SD + i];
  for (int i = tid; i < ng * SD; i += nt) s_goal[i] = a.goal[(size_t)b * ng * SD + i];
  if (do_dyn)
    for (int i = tid; i < n * 2; i += nt) s_act[i] = clampf(a.action[(size_t)b * n * 2 + i], -1.0f, 1.0f);  // env/base.py:84-86
  for (int i = tid; i < no * ostride; i += nt) s_obst[i] = a.obst[(size_t)b * no * ostride + i];
  if (a.hits != nullptr)
    for (int i = tid; i < n * kk * 2; i += nt) s_hpre[i] = a.hits[(size_t)b * n * kk * 2 + i];
  __syncthreads();

  // ---- phase 1a: dynamics + features (thread per agent), goal features (second wave) ----------------
  if (tid < n) {
    const int i = tid;
    const float* x = s_agent + i * SD;
    float* nx = s_next + i * SD;
    if (do_dyn) {
      const float u0 = s_act[i * 2], u1 = s_act[i * 2 + 1];
      const float dt = c.dt, A = c.area_size;
      if constexpr (SD == 5) {  // lidar_bicycle_target.py:95-107
        float theta = atan2f(x[3], x[2]);
        float theta_next = theta + x[4] * u0 * dt * 10.0f;
        nx[0] = clampf(x[0] + x[4] * cosf(theta) * dt, 0.0f, A);
        nx[1] = clampf(x[1] + x[4] * sinf(theta) * dt, 0.0f, A);
        nx[2] = clampf(cosf(theta_next), -1.0f, 1.0f);
        nx[3] = clampf(sinf(theta_next), -1.0f, 1.0f);
        nx[SD - 1] = clampf(x[SD - 1] + u1 * dt * 10.0f, -0.5f, 0.5f);
      } else {        // lidar_env/base.py:146-149
        const float vl = c.vel_limit;
        nx[0] = clampf(x[2] * dt + x[0], 0.0f, A);
        nx[1] = clampf(x[3] * dt + x[1], 0.0f, c.y_limit);       // = A except MPECorridor / MPEConnectSpread (2 A)
        nx[2] = clampf((u0 * 10.0f) * dt + x[2], -vl, vl);
        nx[3] = clampf((u1 * 10.0f) * dt + x[3], -vl, vl);
      }
      // action penalty term: (||a||)^2
      float an = sqrtf(u0 * u0 + u1 * u1);
      s_an2[i] = an * an;
    } else {
      for (int d = 0; d < SD; ++d) nx[d] = x[d];
    }
    state2feat<SD>(nx, s_fa + i * 4);
  }
  for (int g = tid - 64; g >= 0 && g < ng; g += nt) state2feat<SD>(s_goal + g * SD, s_fg + g * 4);
  if (nt <= 64)
    for (int g = tid; g < ng; g += nt) state2feat<SD>(s_goal + g * SD, s_fg + g * 4);
  // reward goals (landmark2goal: lidar_line.py:131-136, mpe_line.py:124-133, mpe_formation.py:94-98)
  if (do_dyn)
    for (int g = tid; g < nrg; g += nt) {
      float gx, gy;
      if (c.reward_goals == DGPPO_GOALS_NODES) { gx = s_goal[g * SD]; gy = s_goal[g * SD + 1]; }
      else if (c.reward_goals == DGPPO_GOALS_CIRCLE) {
        const float th = ((float)g / (float)n) * 6.28318530717958647692f;
        gx = s_goal[0] + c.comm_radius * cosf(th);
        gy = s_goal[1] + c.comm_radius * sinf(th);
      } else {
        const bool ends = c.reward_goals == DGPPO_GOALS_LINE;
        const float i_f = ends ? (float)g : (float)(g + 1), den = ends ? (float)(n - 1) : (float)(n + 1);
        const float dx = s_goal[SD] - s_goal[0], dy = s_goal[SD + 1] - s_goal[1];
        gx = s_goal[0] + (i_f * dx) / den;
        gy = s_goal[1] + (i_f * dy) / den;
      }
      s_rg[g * 2] = gx; s_rg[g * 2 + 1] = gy;
    }
  __syncthreads();

  // ---- phase 1b: all pairwise distances in parallel (one correctly-rounded sqrt per thread) --------------
  // s_pair: [n*n] agent-agent (pre-step, + 1e6 on the diagonal) | [n*oc] agent-obstacle (pre-step) | [ng*n] goal-agent
  // s_ino : [n*no] start-inside-rectangle flags of the t+1 positions (env/utils.py:117, r = 0)
  {
    const int oc = lidar ? kk : no;
    const int n_aa = do_dyn ? n * n : 0, n_ao = do_dyn ? n * oc : 0, n_ga = (do_dyn && spread) ? nrg * n : (do_dyn ? nrg : 0);
    const int n_in = do_sense ? n * no : 0;
    for (int idx = tid; idx < n_aa + n_ao + n_ga + n_in; idx += nt) {
      if (idx < n_aa) {
        const int i = idx / n, j = idx - i * n;
        float dx = s_agent[i * SD] - s_agent[j * SD], dy = s_agent[i * SD + 1] - s_agent[j * SD + 1];
        s_pair[idx] = sqrtf(dx * dx + dy * dy) + ((j == i) ? 1e6f : 0.0f);
      } else if (idx < n_aa + n_ao) {
        const int q = idx - n_aa, i = q / oc, m = q - i * oc;
        float dx, dy;
        if (lidar) { dx = s_hpre[(i * kk + m) * 2] - s_agent[i * SD]; dy = s_hpre[(i * kk + m) * 2 + 1] - s_agent[i * SD + 1]; }
        else { dx = s_agent[i * SD] - s_obst[m * SD]; dy = s_agent[i * SD + 1] - s_obst[m * SD + 1]; }
        s_pair[idx] = sqrtf(dx * dx + dy * dy);
      } else if (idx < n_aa + n_ao + n_ga) {
        const int q = idx - n_aa - n_ao;
        int g, j;
        if (spread) { g = q / n; j = q - g * n; } else { g = q; j = q; }   // each goal finds the nearest agent / paired goal
        float dx = s_rg[g * 2] - s_agent[j * SD], dy = s_rg[g * 2 + 1] - s_agent[j * SD + 1];
        s_pair[idx] = sqrtf(dx * dx + dy * dy);
      } else {
        const int q = idx - n_aa - n_ao - n_ga, i = q / no, o = q - i * no;
        s_ino[q] = rect_inside(s_obst + o * DGPPO_RECT_STRIDE, s_next[i * SD], s_next[i * SD + 1], 0.0f) ? 1.0f : 0.0f;
      }
    }
    __syncthreads();
    // ---- phase 1c: per-agent reductions (min is order independent, NaN propagating), cost, reward terms ----
    if (tid < n) {
      const int i = tid;
      float is_in = 0.0f;
      if (do_sense)
        for (int o = 0; o < no; ++o) is_in = fmaxf(is_in, s_ino[i * no + o]);
      s_isin[i] = is_in;
      if (do_dyn) {  // cost on the PRE-step graph (lidar_env/base.py:180-207, mpe/base.py:164-191)
        float md = s_pair[i * n];
        for (int j = 1; j < n; ++j) md = nanmin(md, s_pair[i * n + j]);
        const float agent_cost = c.two_car_radius - md;
        float obs_cost = 0.0f;
        if (no > 0) {
          float mo = s_pair[n_aa + i * oc];
          for (int m = 1; m < oc; ++m) mo = nanmin(mo, s_pair[n_aa + i * oc + m]);
          obs_cost = (lidar ? c.car_radius : c.car_plus_obs) - mo;
        }
        float c0 = (agent_cost <= 0.0f) ? agent_cost - 0.5f : agent_cost + 0.5f;
        float c1 = (obs_cost <= 0.0f) ? obs_cost - 0.5f : obs_cost + 0.5f;
        if (lidar || c.kind == DGPPO_ENV_MPE_CONNECT_SPREAD) { c0 = clampf_nan(c0, -1.0f, 1.0f); c1 = clampf_nan(c1, -1.0f, 1.0f); }
        else { c0 = fmaxf(c0, -1.0f); c1 = fmaxf(c1, -1.0f); }   // mpe/base.py:189 clips only from below
        a.cost[((size_t)b * n + i) * c.n_cost] = c0;
        a.cost[((size_t)b * n + i) * c.n_cost + 1] = c1;
        s_md[i] = md;
      }
    }
    if (do_dyn) {
      for (int g = tid - 64; g >= 0 && g < nrg; g += nt) {
        float d2g = s_pair[n_aa + n_ao + (spread ? g * n : g)];
        if (spread) for (int j = 1; j < n; ++j) d2g = nanmin(d2g, s_pair[n_aa + n_ao + g * n + j]);
        s_d2g[g] = d2g;
      }
      if (nt <= 64)
        for (int g = tid; g < nrg; g += nt) {
          float d2g = s_pair[n_aa + n_ao + (spread ? g * n : g)];
          if (spread) for (int j = 1; j < n; ++j) d2g = nanmin(d2g, s_pair[n_aa + n_ao + g * n + j]);
          s_d2g[g] = d2g;
        }
    }
  }
  __syncthreads();

  if (do_dyn && c.n_cost == 3 && tid < n) {
    // connectivity (mpe_connect_spread.py:115-117): the largest nearest-neighbour distance against connect_radius, the same
    // value for every agent; NaN propagates like jnp.max
    float w = s_md[0] - c.connect_radius;
    for (int j = 1; j < n; ++j) {
      const float v = s_md[j] - c.connect_radius;
      w = (w != w || v != v) ? __builtin_nanf("") : fmaxf(w, v);
    }
    float c2 = (w <= 0.0f) ? w - 0.5f : w + 0.5f;
    a.cost[((size_t)b * n + tid) * 3 + 2] = clampf_nan(c2, -1.0f, 1.0f);
  }
  if (do_dyn && tid == 0) {
    float s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    for (int g = 0; g < nrg; ++g) s1 = (g == 0) ? s_d2g[0] : s1 + s_d2g[g];
    for (int g = 0; g < nrg; ++g) {
      float ind = (s_d2g[g] > c.dist2goal) ? 1.0f : 0.0f;
      s2 = (g == 0) ? ind : s2 + ind;
    }
    for (int i = 0; i < n; ++i) s3 = (i == 0) ? s_an2[0] : s3 + s_an2[i];
    float r = 0.0f;
    r = r - (s1 / (float)nrg) * 0.01f;
    r = r - (s2 / (float)nrg) * 0.001f;
    r = r - (s3 / (float)n) * 0.0001f;
    a.reward[b] = r;
  }

  // ---- phase 2: ray fan x rectangles x 4 segments (thread per (agent, ray)) ------------------
  // Exactly the reference's arithmetic (obstacle.py:97-105), evaluated without the two IEEE divisions per test:
  //   valid = (0 <= num_a/det <= 1) && (0 <= num_b/det <= 1)  is decided from signs and magnitudes
  //     fl(q) >= 0  <=>  q >= 0   (no quotient of these operands can underflow to -0: |num| is 0 or >= ~1e-22, |det| <= 1e7)
  //     fl(q) <= 1  <=>  q <= 1   (num > det > 0 implies q >= 1 + 2^-24 + eps, which rounds above 1)
  //   and only a valid segment needs alpha = num_a/det (one correctly rounded division).  det == 0 (sign(det) = 0 ->
  //   x/0 -> 0*inf = NaN in the reference) takes the literal slow path so the NaN semantics are preserved bit for bit.
  if (do_sense) {
    const float sr = c.comm_radius;
    // per-segment constants, once per env: P[m] and the edge vector P[m-1] - P[m]
    for (int q = tid; q < no * 4; q += nt) {
      const int o = q >> 2, m = q & 3, mm = (m + 3) & 3;
      const float* P = s_obst + o * DGPPO_RECT_STRIDE + 8;
      s_seg[q * 4 + 0] = P[2 * m];
      s_seg[q * 4 + 1] = P[2 * m + 1];
      s_seg[q * 4 + 2] = P[2 * mm] - P[2 * m];          // x4 - x3
      s_seg[q * 4 + 3] = P[2 * mm + 1] - P[2 * m + 1];  // y4 - y3
    }
    __syncthreads();
    for (int idx = tid; idx < n * R; idx += nt) {
      const int i = idx / R, r = idx - i * R;
      const float x1 = s_next[i * SD], y1 = s_next[i * SD + 1];
      const float x2 = x1 + a.ray_cos[r] * sr;
      const float y2 = y1 + a.ray_sin[r] * sr;
      const float dx12 = x1 - x2, dy12 = y1 - y2;
      float amin = 0.0f;
      for (int o = 0; o < no; ++o) {
        float ao = 0.0f;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const float4 sg = reinterpret_cast<const float4*>(s_seg)[o * 4 + m];
          const float x3 = sg.x, y3 = sg.y, ex = sg.z, ey = sg.w;
          const float det0 = dx12 * ey - dy12 * ex;
          const float ax = x1 - x3, ay = y1 - y3;
          const float na = ey * ax - ex * ay;
          const float nb = (-dy12) * ax + dx12 * ay;
          float al;
          if (det0 != 0.0f && det0 == det0) {
            const float det = copysignf(fminf(fmaxf(fabsf(det0), 1e-7f), 1e7f), det0);
            const bool pos = det > 0.0f;
            const bool va = (na == 0.0f || (na > 0.0f) == pos) && (pos ? (na <= det) : (na >= det));
            const bool vb = (nb == 0.0f || (nb > 0.0f) == pos) && (pos ? (nb <= det) : (nb >= det));
            al = 1e6f;
            if (va && vb) al = na / det + 0.0f;   // v*alpha + (1-v)*1e6 with v = 1: alpha + 0 (turns -0 into +0)
          } else {  // literal path: det = sign(det0) * clip(|det0|) = 0 (or NaN)
            const float sgn = (det0 > 0.0f) ? 1.0f : ((det0 < 0.0f) ? -1.0f : det0);
            const float det = sgn * fminf(fmaxf(fabsf(det0), 1e-7f), 1e7f);
            const float aq = na / det, bq = nb / det;
            const float v = ((aq <= 1.0f) && (aq >= 0.0f) && (bq <= 1.0f) && (bq >= 0.0f)) ? 1.0f : 0.0f;
            al = v * aq + (1.0f - v) * 1e6f;
          }
          ao = (m == 0) ? al : nanmin(ao, al);
        }
        amin = (o == 0) ? ao : nanmin(amin, ao);
      }
      amin = amin * (1.0f - s_isin[i]);
      s_alpha[idx] = amin;
    }
    __syncthreads();
    // ---- phase 3: stable ascending top-k by rank counting (env/utils.py:132-136) --------------
    for (int idx = tid; idx < n * R; idx += nt) {
      const int i = idx / R, r = idx - i * R;
      const float ar = s_alpha[idx];
      const bool nr = (ar != ar);
      int rank = 0;
      for (int j = 0; j < R; ++j) {
        const float aj = s_alpha[i * R + j];
        const bool nj = (aj != aj);
        bool less;
        if (nj || nr) less = (nj == nr) ? (j < r) : nr;  // NaNs sort last, in index order
        else less = (aj < ar) || (aj == ar && j < r);
        rank += less ? 1 : 0;
      }
      if (rank < k) {
        const float x1 = s_next[i * SD], y1 = s_next[i * SD + 1];
        const float x2 = x1 + a.ray_cos[r] * sr;
        const float y2 = y1 + a.ray_sin[r] * sr;
        s_hnext[(i * k + rank) * 2] = x1 + (x2 - x1) * ar;
        s_hnext[(i * k + rank) * 2 + 1] = y1 + (y2 - y1) * ar;
      }
    }
    __syncthreads();
  } else if (lidar && no > 0) {
    // MODE_GRAPH: hits are given
    for (int i = tid; i < n * kk * 2; i += nt) s_hnext[i] = s_hpre[i];
    __syncthreads();
  }

  // ---- phase 4: compact outputs ---------------------------------------------------------------
  if (a.next_agent != nullptr)
    for (int i = tid; i < n * SD; i += nt) a.next_agent[(size_t)b * n * SD + i] = s_next[i];
  if (a.next_hits != nullptr && lidar && no > 0)
    for (int i = tid; i < n * kk * 2; i += nt) a.next_hits[(size_t)b * n * kk * 2 + i] = s_hnext[i];

  if (!a.has_graph) return;
  // ---- phase 5: padded GraphsTuple of the t+1 state --------------------------------------------
  const int N = n + ng + n_on + 1, pad = N - 1;
  const int gslots = spread ? ng : 1;
  const int oslots = lidar ? kk : no;
  const int E = n * (n + gslots + oslots);
  {
    float* nodes = a.g.nodes + (size_t)b * N * ND;
    for (int idx = tid; idx < N * ND; idx += nt) {
      const int node = idx / ND, col = idx - node * ND;
      float v = 0.0f;
      if (node < n) v = (col < SD) ? s_next[node * SD + col] : ((col == SD + 2) ? 1.0f : 0.0f);
      else if (node < n + ng) v = (col < SD) ? s_goal[(node - n) * SD + col] : ((col == SD + 1) ? 1.0f : 0.0f);
      else if (node < pad) {
        const int q = node - n - ng;
        if (lidar) v = (col < 2) ? s_hnext[q * 2 + col] : ((col == SD) ? 1.0f : 0.0f);
        else v = (col < SD) ? s_obst[q * SD + col] : ((col == SD) ? 1.0f : 0.0f);
      }
      nodes[idx] = v;
    }
    float* states = a.g.states + (size_t)b * N * SD;
    for (int idx = tid; idx < N * SD; idx += nt) {
      const int node = idx / SD, col = idx - node * SD;
      float v = -1.0f;  // pad row, graph.py:217
      if (node < n) v = s_next[node * SD + col];
      else if (node < n + ng) v = s_goal[(node - n) * SD + col];
      else if (node < pad) {
        const int q = node - n - ng;
        if (lidar) v = (col < 2) ? s_hnext[q * 2 + col] : 0.0f;
        else v = s_obst[q * SD + col];
      }
      states[idx] = v;
    }
    int32_t* nty = a.g.node_type + (size_t)b * N;
    for (int node = tid; node < N; node += nt) nty[node] = (node < n) ? 0 : ((node < n + ng) ? 1 : ((node < pad) ? 2 : -1));
    if (tid == 0) {
      a.g.n_node[b] = N;
      a.g.n_edge[b] = E;
    }
    float4* edges = reinterpret_cast<float4*>(a.g.edges) + (size_t)b * E;
    int32_t* recv = a.g.receivers + (size_t)b * E;
    int32_t* send = a.g.senders + (size_t)b * E;
    for (int e = tid; e < E; e += nt) {
      int i, sender;
      bool mask;
      float4 f;
      if (e < n * n) {  // agent-agent block, lidar_spread.py:59-67
        i = e / n;
        const int j = e - i * n;
        const float* fi = s_fa + i * 4;
        const float* fj = s_fa + j * 4;
        f = make_float4(fi[0] - fj[0], fi[1] - fj[1], fi[2] - fj[2], fi[3] - fj[3]);
        float dx = s_next[i * SD] - s_next[j * SD], dy = s_next[i * SD + 1] - s_next[j * SD + 1];
        float d = sqrtf(dx * dx + dy * dy) + ((i == j) ? c.eye_offset : 0.0f);
        mask = d < c.comm_radius;
        sender = j;
      } else if (e < n * n + n * gslots) {  // agent-goal
        const int e2 = e - n * n;
        int g;
        if (spread) { i = e2 / ng; g = e2 - i * ng; } else { i = e2; g = e2; }
        const float* fi = s_fa + i * 4;
        const float* fg = s_fg + g * 4;
        f = make_float4(fi[0] - fg[0], fi[1] - fg[1], fi[2] - fg[2], fi[3] - fg[3]);
        mask = true;
        sender = n + g;
      } else {  // agent-obstacle
        const int e3 = e - n * n - n * gslots;
        i = e3 / oslots;
        const int m = e3 - i * oslots;
        if (lidar) {  // lidar_spread.py:79-94
          float lx = s_next[i * SD] - s_hnext[(i * kk + m) * 2];
          float ly = s_next[i * SD + 1] - s_hnext[(i * kk + m) * 2 + 1];
          f = make_float4(lx, ly, 0.0f, 0.0f);
          mask = sqrtf(lx * lx + ly * ly) < c.lidar_mask_radius;
          sender = n + ng + i * kk + m;
        } else {      // mpe_spread.py:73-79
          const float* xi = s_next + i * SD;
          const float* xo = s_obst + m * SD;
          f = make_float4(xi[0] - xo[0], xi[1] - xo[1], xi[2] - xo[2], xi[3] - xo[3]);
          float dx = xi[0] - xo[0], dy = xi[1] - xo[1];
          mask = sqrtf(dx * dx + dy * dy) < c.obs_mask_radius;    // comm_radius, or 100 x for Corridor / ConnectSpread
          sender = n + ng + m;
        }
      }
      edges[e] = f;
      recv[e] = mask ? i : pad;
      send[e] = mask ? sender : pad;
    }
  }
}


// =====================================================================================================================
// Specialised LiDAR kernel for n_rays == 32 (double-integrator or bicycle, Spread or Target goal topology): same
// outputs, bit for bit, as env_step_kernel above; organised around latency (every dependent LDS / global round trip costs
// 100-2000 cycles, arithmetic is cheap):
//   * the lane's ray direction lives in registers for the whole kernel; one half-wave per agent;
//   * all n_obs*4 segment tests of a ray are straight-line code: validity from signs/magnitudes (exact, see above), one
//     pipelined correctly-rounded division per segment, the det == 0 / NaN case on a rare literal slow path;
//   * stable top-k from the agent's 32 keys held in registers (one LDS round trip instead of 32 dependent ones);
//   * node / state rows are composed in LDS and streamed out with unit-stride stores; edges as float4.
// =====================================================================================================================
#ifdef DGPPO_STAMPS
__device__ unsigned long long g_stamps[32];
#define STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int32_t dgppo_debug_stamps(unsigned long long* out) {
  return (int32_t)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 32);
}
#else
#define STAMP(i)
#endif

// x / d for 0 <= x < 2^16 and 1 <= d < 2^16 with rcp = ceil(2^32 / d): one v_mul_hi_u32 instead of the ~35-instruction
// integer division sequence (exact: the error term x * (rcp*d - 2^32) stays below 2^32).
__device__ inline int fdiv(int x, uint32_t rcp) { return (int)__umulhi((uint32_t)x, rcp); }
static inline uint32_t fdiv_rcp(int d) { return d <= 1 ? 0u : (uint32_t)((0x100000000ull + (uint64_t)d - 1) / (uint64_t)d); }

// workgroup barrier that waits for LDS traffic only (no vmcnt wait: outstanding global stores keep draining)
#define LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
template <int SD, bool SPREAD, int NT>
__global__ void __launch_bounds__(NT) __attribute__((amdgpu_num_sgpr(80))) lidar_step_kernel(StepArgs a) {
  // requires n_rays == 32 (one half-wave per agent) — the host falls back to env_step_kernel otherwise
  extern __shared__ float smem[];
  const dgppo_env_cfg& c = a.cfg;
  const int b = blockIdx.x, tid = threadIdx.x;
  constexpr int nt = NT;
  constexpr int R = 32;
  constexpr int GO = (NT > 64) ? 64 : 0;   // first thread of the "goal" work (second wave when there is one)
  const int n = c.n_agents, ng = n, no = c.n_obs, k = c.top_k;
  constexpr int ND = SD + 3;
  const int NR = n * R;
  // index divisions by n, k, n_obs, 4*n_obs through host-precomputed reciprocals (d == 1: identity)
#define DIVN(x) (n == 1 ? (x) : fdiv((x), a.rcp_n))
#define DIVK(x) (k == 1 ? (x) : fdiv((x), a.rcp_k))
#define DIVNO(x) (no == 1 ? (x) : fdiv((x), a.rcp_no))
#define DIVNO4(x) fdiv((x), a.rcp_no4)
  // ---- LDS carve (float4-aligned blocks first) ----
  float4* s_seg = reinterpret_cast<float4*>(smem);                 // no*4: x3, y3, ex, ey
  float4* s_as = s_seg + no * 4;                                   // n*no*4: per (agent, segment) ax, ay, na
  float4* s_row = s_as + n * no * 4;                                    // NR/4: alpha keys, 32 per agent (read back as float4)
  float* s_fa = reinterpret_cast<float*>(s_row + NR / 4);          // n*4
  float* s_fg = s_fa + n * 4;                                      // n*4
  float* s_obst = s_fg + n * 4;                                    // no*16
  float* s_agent = s_obst + no * 16;                               // n*SD
  float* s_next = s_agent + n * SD;                                // n*SD
  float* s_goal = s_next + n * SD;                                 // n*SD
  float* s_act = s_goal + n * SD;                                  // n*2
  float* s_hpre = s_act + n * 2;                                   // n*k*2
  float* s_hnext = s_hpre + n * k * 2;                             // n*k*2
  float* s_pair = s_hnext + n * k * 2;                             // n*n + n*k + n*n
  float* s_ino = s_pair + 2 * n * n + n * k;                       // n*no
  float* s_d2g = s_ino + n * no;                                   // n
  float* s_an2 = s_d2g + n;                                        // n
  float* s_outn = s_an2 + n;                                       // N*ND node rows
  float* s_outs = s_outn + (2 * n + n * k + 1) * ND;               // N*SD state rows
  const bool do_dyn = (a.mode == MODE_STEP);
  const bool do_sense = (a.mode != MODE_GRAPH);
  const float sr = c.comm_radius;
  // this lane's ray, for the whole kernel (no tables in materialise-only mode)
  const float cr = do_sense ? a.ray_cos[tid & 31] : 0.0f, sn = do_sense ? a.ray_sin[tid & 31] : 0.0f;
  STAMP(0);
  // ---- P0: stage inputs ----
  for (int i = tid; i < n * SD; i += nt) { s_agent[i] = a.agent[(size_t)b * n * SD + i]; s_goal[i] = a.goal[(size_t)b * n * SD + i]; }
  if (do_dyn) for (int i = tid; i < n * 2; i += nt) s_act[i] = clampf(a.action[(size_t)b * n * 2 + i], -1.0f, 1.0f);
  for (int i = tid; i < no * 16; i += nt) s_obst[i] = a.obst[(size_t)b * no * 16 + i];
  if (a.hits != nullptr) for (int i = tid; i < n * k * 2; i += nt) s_hpre[i] = a.hits[(size_t)b * n * k * 2 + i];
  LDS_BARRIER();   // threads exchange data through LDS only; __syncthreads() would also wait for the global stores
  STAMP(1);
  // ---- P1a: per-segment constants, bounding circles, dynamics, features (disjoint lanes) ----
  if (do_sense) {
    for (int q = tid; q < no * 4; q += nt) {
      const int o = q >> 2, m = q & 3, mm = (m + 3) & 3;
      const float* P = s_obst + o * 16 + 8;
      s_seg[q] = make_float4(P[2 * m], P[2 * m + 1], P[2 * mm] - P[2 * m], P[2 * mm + 1] - P[2 * m + 1]);
    }
  }
  if (tid < n) {
    const int i = tid;
    const float* x = s_agent + i * SD;
    float* nx = s_next + i * SD;
    if (do_dyn) {
      const float u0 = s_act[i * 2], u1 = s_act[i * 2 + 1];
      const float dt = c.dt, A = c.area_size;
      if constexpr (SD == 5) {
        float theta = atan2f(x[3], x[2]);
        float theta_next = theta + x[4] * u0 * dt * 10.0f;
        nx[0] = clampf(x[0] + x[4] * cosf(theta) * dt, 0.0f, A);
        nx[1] = clampf(x[1] + x[4] * sinf(theta) * dt, 0.0f, A);
        nx[2] = clampf(cosf(theta_next), -1.0f, 1.0f);
        nx[3] = clampf(sinf(theta_next), -1.0f, 1.0f);
        nx[SD - 1] = clampf(x[SD - 1] + u1 * dt * 10.0f, -0.5f, 0.5f);
      } else {
        const float vl = c.vel_limit;
        nx[0] = clampf(x[2] * dt + x[0], 0.0f, A);
        nx[1] = clampf(x[3] * dt + x[1], 0.0f, A);
        nx[2] = clampf((u0 * 10.0f) * dt + x[2], -vl, vl);
        nx[3] = clampf((u1 * 10.0f) * dt + x[3], -vl, vl);
      }
      const float an = sqrtf(u0 * u0 + u1 * u1);
      s_an2[i] = an * an;
    } else {
      for (int d = 0; d < SD; ++d) nx[d] = x[d];
    }
    state2feat<SD>(nx, s_fa + i * 4);
  }
  if (tid >= GO && tid < GO + n) state2feat<SD>(s_goal + (tid - GO) * SD, s_fg + (tid - GO) * 4);
  LDS_BARRIER();   // threads exchange data through LDS only; __syncthreads() would also wait for the global stores
  STAMP(2);
  // ---- P1b: distances (one sqrt per thread) and start-inside flags ----
  {
    const int n_aa = do_dyn ? n * n : 0, n_ao = do_dyn ? n * k : 0, n_ga = do_dyn ? (SPREAD ? n * n : n) : 0;
    const int n_in = do_sense ? n * no : 0, n_as = do_sense ? n * no * 4 : 0;
    for (int idx = tid; idx < n_aa + n_ao + n_ga + n_in + n_as; idx += nt) {
      if (idx < n_aa) {
        const int i = DIVN(idx), j = idx - i * n;
        const float dx = s_agent[i * SD] - s_agent[j * SD], dy = s_agent[i * SD + 1] - s_agent[j * SD + 1];
        s_pair[idx] = sqrtf(dx * dx + dy * dy) + ((j == i) ? 1e6f : 0.0f);
      } else if (idx < n_aa + n_ao) {
        const int q = idx - n_aa, i = DIVK(q);
        const float dx = s_hpre[q * 2] - s_agent[i * SD], dy = s_hpre[q * 2 + 1] - s_agent[i * SD + 1];
        s_pair[idx] = sqrtf(dx * dx + dy * dy);
      } else if (idx < n_aa + n_ao + n_ga) {
        const int q = idx - n_aa - n_ao;
        const int g = SPREAD ? DIVN(q) : q, j = SPREAD ? q - g * n : q;
        const float dx = s_goal[g * SD] - s_agent[j * SD], dy = s_goal[g * SD + 1] - s_agent[j * SD + 1];
        s_pair[idx] = sqrtf(dx * dx + dy * dy);
      } else if (idx < n_aa + n_ao + n_ga + n_in) {
        const int q = idx - n_aa - n_ao - n_ga, i = DIVNO(q), o = q - i * no;
        s_ino[q] = rect_inside(s_obst + o * 16, s_next[i * SD], s_next[i * SD + 1], 0.0f) ? 1.0f : 0.0f;
      } else {  // ray-independent part of the segment test: (x1-x3, y1-y3, (y4-y3)(x1-x3) - (x4-x3)(y1-y3))
        const int q = idx - n_aa - n_ao - n_ga - n_in, i = DIVNO4(q), sgi = q - i * (no * 4);
        const float4 sg = s_seg[sgi];
        const float ax = s_next[i * SD] - sg.x, ay = s_next[i * SD + 1] - sg.y;
        s_as[q] = make_float4(ax, ay, sg.w * ax - sg.z * ay, 0.0f);
      }
    }
  }
  LDS_BARRIER();   // threads exchange data through LDS only; __syncthreads() would also wait for the global stores
  STAMP(3);
  // ---- P1c: per-agent reductions (min is order independent), cost, reward terms ----
  if (do_dyn) {
    if (tid < n) {
      const int i = tid;
      float md = s_pair[i * n];
#pragma unroll 8
      for (int j = 1; j < n; ++j) md = nanmin(md, s_pair[i * n + j]);
      float mo = s_pair[n * n + i * k];
#pragma unroll 8
      for (int m = 1; m < k; ++m) mo = nanmin(mo, s_pair[n * n + i * k + m]);
      const float agent_cost = c.two_car_radius - md, obs_cost = c.car_radius - mo;
      const float c0 = (agent_cost <= 0.0f) ? agent_cost - 0.5f : agent_cost + 0.5f;
      const float c1 = (obs_cost <= 0.0f) ? obs_cost - 0.5f : obs_cost + 0.5f;
      reinterpret_cast<float2*>(a.cost)[(size_t)b * n + i] = make_float2(clampf_nan(c0, -1.0f, 1.0f), clampf_nan(c1, -1.0f, 1.0f));
    }
    if (tid >= GO && tid < GO + n) {
      const int g = tid - GO, base = n * n + n * k;
      float d2g = s_pair[base + (SPREAD ? g * n : g)];
      if (SPREAD) {
#pragma unroll 8
        for (int j = 1; j < n; ++j) d2g = nanmin(d2g, s_pair[base + g * n + j]);
      }
      s_d2g[g] = d2g;
    }
  }
  STAMP(4);
  // ---- P2: all n_obs*4 segment tests of (agent, this lane's ray) (reference arithmetic, obstacle.py:97-105).  Per
  //      obstacle the 4 segments are unrolled: one batch of LDS reads, validity without divisions (see the generic
  //      kernel), one division per segment only when some lane of the wave hits it.  det == 0 / NaN (ray parallel to an
  //      edge: the reference then yields NaN or 1e6 through 0*inf) is only FLAGGED in the loop; a flagged lane redoes
  //      all of its segments afterwards with the literal reference arithmetic (identical results for the segments
  //      the fast path handles, which is what makes the two paths interchangeable).
  if (do_sense) {
    for (int base = 0; base < NR; base += nt) {
      const int idx = base + tid;
      if (idx < NR) {
        const int i = idx >> 5;
        const float x1 = s_next[i * SD], y1 = s_next[i * SD + 1];
        const float x2 = x1 + cr * sr, y2 = y1 + sn * sr;
        const float dx12 = x1 - x2, dy12 = y1 - y2, ndy12 = -dy12;
        float amin = 1e6f, is_in = 0.0f;
        bool bad = false;
        for (int o = 0; o < no; ++o) is_in = fmaxf(is_in, s_ino[i * no + o]);
        for (int o = 0; o < no; ++o) {
          float4 sg[4], as[4];
#pragma unroll
          for (int m = 0; m < 4; ++m) { sg[m] = s_seg[o * 4 + m]; as[m] = s_as[(i * no + o) * 4 + m]; }
          float naf[4], adet[4];
          bool valid[4];
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            const float det0 = dx12 * sg[m].w - dy12 * sg[m].z;
            const float nb = ndy12 * as[m].x + dx12 * as[m].y;
            // flip both numerators by the sign of det: (na/det, nb/det) == (na'/|det|, nb'/|det|), exactly
            const uint32_t sb = __float_as_uint(det0) & 0x80000000u;
            naf[m] = __uint_as_float(__float_as_uint(as[m].z) ^ sb);
            const float nbf = __uint_as_float(__float_as_uint(nb) ^ sb);
            adet[m] = __builtin_amdgcn_fmed3f(fabsf(det0), 1e-7f, 1e7f);      // clip(|det|, 1e-7, 1e7)
            float mn, mx;                  // raw min / max: operands are never signalling NaNs that would need quieting
            asm("v_min_f32 %0, %1, %2" : "=v"(mn) : "v"(naf[m]), "v"(nbf));
            asm("v_max_f32 %0, %1, %2" : "=v"(mx) : "v"(naf[m]), "v"(nbf));
            // 0 <= q <= 1 for both quotients  <=>  0 <= min(na', nb') and max(na', nb') <= |det|   (-0 >= 0 holds, like -0/d >= 0)
            valid[m] = (mn >= 0.0f) && (mx <= adet[m]);
            bad = bad || !(det0 != 0.0f);                                      // zero or NaN
          }
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            if (__builtin_amdgcn_ballot_w64(valid[m]) != 0ull) {              // wave-uniform: skip the division when no lane hits
              const float qa = naf[m] / adet[m] + 0.0f;                       // v*alpha + (1-v)*1e6 with v = 1 (turns -0 into +0)
              const float al = valid[m] ? qa : 1e6f;
              asm("v_min_f32 %0, %1, %2" : "=v"(amin) : "v"(amin), "v"(al));
            }
          }
        }
        float ar = amin;
        if (bad) {                         // literal reference arithmetic for every segment of this lane
          float lmin = 1e6f;
          bool any_nan = false;
          for (int q = 0; q < no * 4; ++q) {
            const float4 sgq = s_seg[q];
            const float4 asq = s_as[i * no * 4 + q];
            const float det0 = dx12 * sgq.w - dy12 * sgq.z;
            const float nb = ndy12 * asq.x + dx12 * asq.y;
            const float sgn = (det0 > 0.0f) ? 1.0f : ((det0 < 0.0f) ? -1.0f : det0);
            const float dz = sgn * fminf(fmaxf(fabsf(det0), 1e-7f), 1e7f);
            const float aq = asq.z / dz, bq = nb / dz;
            const float v = ((aq <= 1.0f) && (aq >= 0.0f) && (bq <= 1.0f) && (bq >= 0.0f)) ? 1.0f : 0.0f;
            const float al = v * aq + (1.0f - v) * 1e6f;
            any_nan = any_nan || (al != al);
            lmin = fminf(lmin, al);
          }
          ar = any_nan ? __builtin_nanf("") : lmin;
        }
        ar = ar * (1.0f - is_in);
        // sort key: float bits (alphas are >= +0), NaN -> max
        reinterpret_cast<uint32_t*>(s_row)[idx] = (ar != ar) ? 0xFFFFFFFFu : __float_as_uint(ar);
      }
    }
  }
  LDS_BARRIER();   // threads exchange data through LDS only; __syncthreads() would also wait for the global stores
  STAMP(5);
  if (do_dyn && tid == 0) {
    float s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
#pragma unroll 8
    for (int g = 0; g < ng; ++g) s1 = (g == 0) ? s_d2g[0] : s1 + s_d2g[g];
#pragma unroll 8
    for (int g = 0; g < ng; ++g) { const float ind = (s_d2g[g] > c.dist2goal) ? 1.0f : 0.0f; s2 = (g == 0) ? ind : s2 + ind; }
    for (int i = 0; i < n; ++i) s3 = (i == 0) ? s_an2[0] : s3 + s_an2[i];
    float r = 0.0f;
    r = r - (s1 / (float)ng) * 0.01f;
    r = r - (s2 / (float)ng) * 0.001f;
    r = r - (s3 / (float)n) * 0.0001f;
    a.reward[b] = r;
  }
  STAMP(6);
  // ---- P3: stable ascending top-k (env/utils.py:132-136): the agent's 32 keys in registers, one LDS round trip ----
  if (do_sense) {
    for (int base = 0; base < NR; base += nt) {
      const int idx = base + tid;
      if (idx < NR) {
        const int i = idx >> 5, r = idx & 31;
        uint32_t key[32];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const float4 v = s_row[i * 8 + q];
          key[4 * q] = __float_as_uint(v.x); key[4 * q + 1] = __float_as_uint(v.y);
          key[4 * q + 2] = __float_as_uint(v.z); key[4 * q + 3] = __float_as_uint(v.w);
        }
        const uint32_t kr = reinterpret_cast<uint32_t*>(s_row)[idx];
        int rank = 0;
#pragma unroll
        for (int j = 0; j < 32; ++j) rank += (key[j] < kr || (key[j] == kr && j < r)) ? 1 : 0;
        if (rank < k) {
          const float ar = (kr == 0xFFFFFFFFu) ? __builtin_nanf("") : __uint_as_float(kr);
          const float x1 = s_next[i * SD], y1 = s_next[i * SD + 1];
          const float x2 = x1 + cr * sr, y2 = y1 + sn * sr;
          s_hnext[(i * k + rank) * 2] = x1 + (x2 - x1) * ar;
          s_hnext[(i * k + rank) * 2 + 1] = y1 + (y2 - y1) * ar;
        }
      }
    }
  } else {
    for (int i = tid; i < n * k * 2; i += nt) s_hnext[i] = s_hpre[i];
  }
  LDS_BARRIER();   // threads exchange data through LDS only; __syncthreads() would also wait for the global stores
  STAMP(7);
  // ---- P4: compact outputs ----
  if (a.next_agent != nullptr) for (int i = tid; i < n * SD; i += nt) a.next_agent[(size_t)b * n * SD + i] = s_next[i];
  if (a.next_hits != nullptr) for (int i = tid; i < n * k * 2; i += nt) a.next_hits[(size_t)b * n * k * 2 + i] = s_hnext[i];
  STAMP(8);
  if (!a.has_graph) return;
  // ---- P5: padded GraphsTuple: node / state rows composed in LDS, streamed out with unit stride ----
  const int n_on = n * k, N = 2 * n + n_on + 1, pad = N - 1;
  const int gslots = SPREAD ? n : 1;
  const int E = n * (n + gslots + k);
  for (int node = tid; node < N; node += nt) {
    float* row = s_outn + node * ND;
    float* srow = s_outs + node * SD;
#pragma unroll
    for (int col = 0; col < ND; ++col) row[col] = 0.0f;
    if (node < n) {
#pragma unroll
      for (int d = 0; d < SD; ++d) { const float v = s_next[node * SD + d]; row[d] = v; srow[d] = v; }
      row[SD + 2] = 1.0f;
    } else if (node < 2 * n) {
#pragma unroll
      for (int d = 0; d < SD; ++d) { const float v = s_goal[(node - n) * SD + d]; row[d] = v; srow[d] = v; }
      row[SD + 1] = 1.0f;
    } else if (node < pad) {
      const float hx = s_hnext[(node - 2 * n) * 2], hy = s_hnext[(node - 2 * n) * 2 + 1];
      row[0] = hx; row[1] = hy; row[SD] = 1.0f;
      srow[0] = hx; srow[1] = hy;
#pragma unroll
      for (int d = 2; d < SD; ++d) srow[d] = 0.0f;
    } else {
#pragma unroll
      for (int d = 0; d < SD; ++d) srow[d] = -1.0f;   // pad row, graph.py:217
    }
  }
  // edges need no staging: one float4 + two ints per thread
  {
    float4* edges = reinterpret_cast<float4*>(a.g.edges) + (size_t)b * E;
    int32_t* recv = a.g.receivers + (size_t)b * E;
    int32_t* send = a.g.senders + (size_t)b * E;
    const int n_aa = n * n, n_ag = n * gslots;
    for (int e = tid; e < E; e += nt) {
      int i, sender;
      bool mask;
      float4 f;
      if (e < n_aa) {
        i = DIVN(e);
        const int j = e - i * n;
        const float* fi = s_fa + i * 4;
        const float* fj = s_fa + j * 4;
        f = make_float4(fi[0] - fj[0], fi[1] - fj[1], fi[2] - fj[2], fi[3] - fj[3]);
        const float dx = s_next[i * SD] - s_next[j * SD], dy = s_next[i * SD + 1] - s_next[j * SD + 1];
        mask = (sqrtf(dx * dx + dy * dy) + ((i == j) ? c.eye_offset : 0.0f)) < c.comm_radius;
        sender = j;
      } else if (e < n_aa + n_ag) {
        const int e2 = e - n_aa;
        int g;
        if (SPREAD) { i = DIVN(e2); g = e2 - i * n; } else { i = e2; g = e2; }
        const float* fi = s_fa + i * 4;
        const float* fg = s_fg + g * 4;
        f = make_float4(fi[0] - fg[0], fi[1] - fg[1], fi[2] - fg[2], fi[3] - fg[3]);
        mask = true;
        sender = n + g;
      } else {
        const int e3 = e - n_aa - n_ag;
        i = DIVK(e3);
        const float lx = s_next[i * SD] - s_hnext[e3 * 2], ly = s_next[i * SD + 1] - s_hnext[e3 * 2 + 1];
        f = make_float4(lx, ly, 0.0f, 0.0f);
        mask = sqrtf(lx * lx + ly * ly) < c.lidar_mask_radius;
        sender = 2 * n + e3;
      }
      edges[e] = f;
      recv[e] = mask ? i : pad;
      send[e] = mask ? sender : pad;
    }
    int32_t* nty = a.g.node_type + (size_t)b * N;
    for (int node = tid; node < N; node += nt) nty[node] = (node < n) ? 0 : ((node < 2 * n) ? 1 : ((node < pad) ? 2 : -1));
    if (tid == 0) { a.g.n_node[b] = N; a.g.n_edge[b] = E; }
  }
  LDS_BARRIER();   // threads exchange data through LDS only; __syncthreads() would also wait for the global stores
  {
    float* nodes = a.g.nodes + (size_t)b * N * ND;
    for (int idx = tid; idx < N * ND; idx += nt) nodes[idx] = s_outn[idx];
    float* states = a.g.states + (size_t)b * N * SD;
    for (int idx = tid; idx < N * SD; idx += nt) states[idx] = s_outs[idx];
  }
  STAMP(9);
}
#undef DIVN
#undef DIVK
#undef DIVNO
#undef DIVNO4

static size_t lidar_smem_bytes(const dgppo_env_cfg& c) {
  const size_t n = c.n_agents, no = c.n_obs, R = 32, k = c.top_k, SD = c.state_dim, NR = n * R;
  const size_t N = 2 * n + n * k + 1;
  size_t w = no * 16 + n * no * 16 + NR + n * 8 + no * 16 + n * SD * 3 + n * 2 + n * k * 4 + (2 * n * n + n * k) + n * no + 2 * n +
             N * (SD + 3) + N * SD;
  return w * 4;
}

static size_t step_smem_bytes(const dgppo_env_cfg& c) {
  const int n = c.n_agents, ng = c.n_goals, no = c.n_obs, SD = c.state_dim;
  const bool lidar = cfg_is_lidar(c);
  const int kk = lidar ? (no > 0 ? c.top_k : 0) : 0;
  size_t fl = (lidar ? (size_t)no * 16 : 0) + (size_t)n * SD * 2 + (size_t)ng * SD + n * 2 + (size_t)no * cfg_obst_stride(c) + (size_t)n * kk * 4 +
              (lidar ? (size_t)n * c.n_rays : 0) + n * 4 + ng * 4 + 3 * (size_t)cfg_reward_goals(c) + 3 * n + (size_t)n * no +
              (size_t)n * n + (size_t)n * (kk > no ? kk : no) + (size_t)cfg_reward_goals(c) * n;
  return fl * sizeof(float);
}

// smallest float s >= 0 whose correctly rounded square root is >= c: for s >= 0, fl(sqrt(s)) < c  <=>  s < sqrt_threshold(c)
// (fl(sqrt) is monotone), so a radius mask can compare the squared distance and skip the root without changing a bit
static float sqrt_threshold(float c) {
  if (!(c > 0.0f)) return 0.0f;
  float s = c * c;
  while (s > 0.0f && sqrtf(s) >= c) s = nextafterf(s, 0.0f);
  while (!(sqrtf(s) >= c)) s = nextafterf(s, INFINITY);
  return s;
}

static int32_t launch_step(const dgppo_env_cfg* cfg, int mode, const float* agent, const float* action, const float* goal,
                           const float* obst, const float* hits, const float* ray_cos, const float* ray_sin,
                           float* next_agent, float* next_hits, float* reward, float* cost, const dgppo_graph_out* gout,
                           int32_t B, void* stream) {
  int32_t rc = dgppo_validate_cfg(cfg);
  if (rc) return rc;
  DGPPO_REQUIRE(B >= 0, "B must be >= 0 (got %d)", B);
  if (B == 0) return 0;
  const bool lidar = cfg_is_lidar(*cfg);
  DGPPO_REQUIRE(agent && goal, "agent/goal must not be NULL");
  DGPPO_REQUIRE(cfg->n_obs == 0 || obst, "obst must not be NULL when n_obs > 0");
  if (mode == MODE_STEP) {
    DGPPO_REQUIRE(action && reward && cost && next_agent, "step needs action, reward, cost and next_agent");
    DGPPO_REQUIRE(!(lidar && cfg->n_obs > 0) || hits, "LiDAR step needs the hit points of the current graph");
  }
  if (mode != MODE_GRAPH && lidar && cfg->n_obs > 0) {
    DGPPO_REQUIRE(ray_cos && ray_sin, "LiDAR sensing needs the ray tables");
    DGPPO_REQUIRE(next_hits || gout, "LiDAR sensing needs next_hits or a graph output");
  }
  if (mode == MODE_GRAPH) {
    DGPPO_REQUIRE(gout, "materialize needs a graph output");
    DGPPO_REQUIRE(!(lidar && cfg->n_obs > 0) || hits, "materialize needs hits for LiDAR envs");
  }
  StepArgs a;
  a.cfg = *cfg;
  a.agent = agent; a.action = action; a.goal = goal; a.obst = obst; a.hits = hits;
  a.ray_cos = ray_cos; a.ray_sin = ray_sin;
  a.next_agent = next_agent; a.next_hits = next_hits; a.reward = reward; a.cost = cost;
  a.has_graph = 0;
  a.g = dgppo_graph_out{};
  if (gout) {
    DGPPO_REQUIRE(gout->nodes && gout->edges && gout->states && gout->receivers && gout->senders && gout->node_type &&
                      gout->n_node && gout->n_edge,
                  "graph output must have all eight pointers set");
    DGPPO_REQUIRE(((uintptr_t)gout->edges & 15) == 0, "graph edges must be 16-byte aligned");
    a.g = *gout;
    a.has_graph = 1;
  }
  a.mode = mode;
  a.B = B;
  a.rcp_n = fdiv_rcp(cfg->n_agents); a.rcp_k = fdiv_rcp(cfg->top_k); a.rcp_no = fdiv_rcp(cfg->n_obs);
  a.rcp_no4 = fdiv_rcp(4 * (cfg->n_obs > 0 ? cfg->n_obs : 1));
  a.thr2_comm = sqrt_threshold(cfg->comm_radius);
  a.thr2_lidar = sqrt_threshold(cfg->lidar_mask_radius);
  const size_t smem = step_smem_bytes(*cfg);
  DGPPO_REQUIRE(smem <= 64 * 1024, "env too large for the per-env LDS stage (%zu bytes)", smem);
  int work = cfg->n_agents * (lidar ? cfg->n_rays : 1);
  int E = cfg_num_edges(*cfg);
  if (gout && E > work) work = E;
  int threads = ((work + 63) / 64) * 64;
  if (threads < 64) threads = 64;
  if (threads > 512) threads = 512;
  hipStream_t s = (hipStream_t)stream;
  const size_t fsmem = lidar ? lidar_smem_bytes(*cfg) : 0;
  if (launch_lidar_wave(a, s)) {
    // wave-per-env kernel for the benchmark topologies (env_wave.hip; same outputs bit for bit)
  } else if (lidar && cfg_is_base_kind(*cfg) && cfg->n_obs > 0 && cfg->n_rays == 32 && fsmem <= 60 * 1024 &&
             !getenv("DGPPO_GENERIC_ENV_KERNEL")) {
    // specialised LiDAR kernel (same outputs bit for bit; see its header)
    const bool spread = cfg_is_spread(*cfg);
    const char* nt_env = getenv("DGPPO_ENV_BLOCK");
    const int ntb = nt_env ? atoi(nt_env) : 128;
    if (ntb == 128) {
      if (cfg->state_dim == 5) hipLaunchKernelGGL((lidar_step_kernel<5, false, 128>), dim3(B), dim3(128), fsmem, s, a);
      else if (spread) hipLaunchKernelGGL((lidar_step_kernel<4, true, 128>), dim3(B), dim3(128), fsmem, s, a);
      else hipLaunchKernelGGL((lidar_step_kernel<4, false, 128>), dim3(B), dim3(128), fsmem, s, a);
    } else if (ntb == 64) {
      if (cfg->state_dim == 5) hipLaunchKernelGGL((lidar_step_kernel<5, false, 64>), dim3(B), dim3(64), fsmem, s, a);
      else if (spread) hipLaunchKernelGGL((lidar_step_kernel<4, true, 64>), dim3(B), dim3(64), fsmem, s, a);
      else hipLaunchKernelGGL((lidar_step_kernel<4, false, 64>), dim3(B), dim3(64), fsmem, s, a);
    } else {
      if (cfg->state_dim == 5) hipLaunchKernelGGL((lidar_step_kernel<5, false, 256>), dim3(B), dim3(256), fsmem, s, a);
      else if (spread) hipLaunchKernelGGL((lidar_step_kernel<4, true, 256>), dim3(B), dim3(256), fsmem, s, a);
      else hipLaunchKernelGGL((lidar_step_kernel<4, false, 256>), dim3(B), dim3(256), fsmem, s, a);
    }
  } else if (cfg->state_dim == 5)
    hipLaunchKernelGGL(env_step_kernel<5>, dim3(B), dim3(threads), smem, s, a);
  else
    hipLaunchKernelGGL(env_step_kernel<4>, dim3(B), dim3(threads), smem, s, a);
  DGPPO_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t dgppo_env_step(const dgppo_env_cfg* cfg, const float* agent, const float* action, const float* goal,
                                  const float* obst, const float* hits, const float* ray_cos, const float* ray_sin,
                                  float* next_agent, float* next_hits, float* reward, float* cost,
                                  const dgppo_graph_out* gout, int32_t B, void* stream) {
  if (!cfg) { dgppo_set_error("cfg is NULL"); return -1; }
  const int mode = action ? MODE_STEP : MODE_SENSE;
  return launch_step(cfg, mode, agent, action, goal, obst, hits, ray_cos, ray_sin, next_agent, next_hits, reward, cost,
                     gout, B, stream);
}

extern "C" int32_t dgppo_graph_materialize(const dgppo_env_cfg* cfg, const float* agent, const float* goal,
                                           const float* obst, const float* hits, const dgppo_graph_out* gout, int32_t B,
                                           void* stream) {
  if (!cfg) { dgppo_set_error("cfg is NULL"); return -1; }
  return launch_step(cfg, MODE_GRAPH, agent, nullptr, goal, obst, hits, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                     gout, B, stream);
}
