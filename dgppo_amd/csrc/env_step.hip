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
#include "common.h"

enum { MODE_STEP = 0, MODE_SENSE = 1, MODE_GRAPH = 2 };

struct StepArgs {
  dgppo_env_cfg cfg;
  const float* agent;
  const float* action;
  const float* goal;
  const float* obst;
  const float* hits;
  const float* ray_cos;
  const float* ray_sin;
  float* next_agent;
  float* next_hits;
  float* reward;
  float* cost;
  dgppo_graph_out g;
  int has_graph;
  int mode;
  int B;
};

// state2feat: lidar_bicycle_target.py:113-118 (identity for the double integrator)
template <int SD>
__device__ inline void state2feat(const float* s, float* f) {
  if constexpr (SD == 5) {
    f[0] = s[0];
    f[1] = s[1];
    f[2] = s[4] * s[2];
    f[3] = s[4] * s[3];
  } else {
    f[0] = s[0]; f[1] = s[1]; f[2] = s[2]; f[3] = s[3];
  }
}

__device__ inline float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

// Rectangle.inside with radius r (obstacle.py:62-72); rec = 16-float record
__device__ inline bool rect_inside(const float* rec, float px, float py, float r) {
  float rel_x = px - rec[0];
  float rel_y = py - rec[1];
  float c = rec[5], s = rec[6];
  float rel_xx = fabsf(rel_x * c + rel_y * s) - rec[2] / 2.0f;
  float rel_yy = fabsf(rel_x * s - rel_y * c) - rec[3] / 2.0f;
  bool is_in_down = (rel_xx < r) && (rel_yy < 0.0f);
  bool is_in_up = (rel_xx < 0.0f) && (rel_yy < r);
  bool is_out_corner = (rel_xx > 0.0f) && (rel_yy > 0.0f);
  bool is_in_circle = sqrtf(rel_xx * rel_xx + rel_yy * rel_yy) < r;
  return is_in_down || is_in_up || (is_out_corner && is_in_circle);
}

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

  float* s_agent = smem;                    // n*SD   state at t
  float* s_next = s_agent + n * SD;         // n*SD   state at t+1
  float* s_goal = s_next + n * SD;          // ng*SD
  float* s_act = s_goal + ng * SD;          // n*2    clipped action
  float* s_obst = s_act + n * 2;            // no*ostride
  float* s_hpre = s_obst + no * ostride;    // n*kk*2 hits of graph_t
  float* s_hnext = s_hpre + n * kk * 2;     // n*kk*2 hits of graph_{t+1}
  float* s_alpha = s_hnext + n * kk * 2;    // n*R
  float* s_fa = s_alpha + (lidar ? n * R : 0);  // n*4   state2feat(next agent)
  float* s_fg = s_fa + n * 4;               // ng*4
  float* s_d2g = s_fg + ng * 4;             // ng
  float* s_an2 = s_d2g + ng;                // n
  float* s_isin = s_an2 + n;                // n (0/1)

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

  // ---- phase 1: dynamics, features, reward terms, cost (thread per agent / goal) -------------
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
        nx[1] = clampf(x[3] * dt + x[1], 0.0f, A);
        nx[2] = clampf((u0 * 10.0f) * dt + x[2], -vl, vl);
        nx[3] = clampf((u1 * 10.0f) * dt + x[3], -vl, vl);
      }
    } else {
      for (int d = 0; d < SD; ++d) nx[d] = x[d];
    }
    state2feat<SD>(nx, s_fa + i * 4);
    // start-inside-obstacle flag for the t+1 position (env/utils.py:117, r = 0)
    bool is_in = false;
    if (do_sense)
      for (int o = 0; o < no; ++o) is_in = is_in || rect_inside(s_obst + o * DGPPO_RECT_STRIDE, nx[0], nx[1], 0.0f);
    s_isin[i] = is_in ? 1.0f : 0.0f;

    if (do_dyn) {
      // ---- cost on the PRE-step graph (lidar_env/base.py:180-207, mpe/base.py:164-191) ----
      const float px = x[0], py = x[1];
      float md = 0.0f;
      for (int j = 0; j < n; ++j) {
        float dx = px - s_agent[j * SD], dy = py - s_agent[j * SD + 1];
        float d = sqrtf(dx * dx + dy * dy) + ((j == i) ? 1e6f : 0.0f);
        md = (j == 0) ? d : nanmin(md, d);
      }
      float agent_cost = c.two_car_radius - md;
      float obs_cost = 0.0f;
      if (no > 0) {
        float mo = 0.0f;
        if (lidar) {
          for (int m = 0; m < kk; ++m) {
            float dx = s_hpre[(i * kk + m) * 2] - px, dy = s_hpre[(i * kk + m) * 2 + 1] - py;
            float d = sqrtf(dx * dx + dy * dy);
            mo = (m == 0) ? d : nanmin(mo, d);
          }
          obs_cost = c.car_radius - mo;
        } else {
          for (int o = 0; o < no; ++o) {
            float dx = px - s_obst[o * SD], dy = py - s_obst[o * SD + 1];
            float d = sqrtf(dx * dx + dy * dy);
            mo = (o == 0) ? d : nanmin(mo, d);
          }
          obs_cost = c.car_plus_obs - mo;
        }
      }
      float c0 = (agent_cost <= 0.0f) ? agent_cost - 0.5f : agent_cost + 0.5f;
      float c1 = (obs_cost <= 0.0f) ? obs_cost - 0.5f : obs_cost + 0.5f;
      if (lidar) {
        c0 = clampf(c0, -1.0f, 1.0f);
        c1 = clampf(c1, -1.0f, 1.0f);
      } else {  // mpe/base.py:189 clips only from below
        c0 = fmaxf(c0, -1.0f);
        c1 = fmaxf(c1, -1.0f);
      }
      a.cost[((size_t)b * n + i) * 2] = c0;
      a.cost[((size_t)b * n + i) * 2 + 1] = c1;
      // action penalty term: (||a||)^2
      float an = sqrtf(s_act[i * 2] * s_act[i * 2] + s_act[i * 2 + 1] * s_act[i * 2 + 1]);
      s_an2[i] = an * an;
    }
  }
  if (tid < ng) {
    const int g = tid;
    state2feat<SD>(s_goal + g * SD, s_fg + g * 4);
    if (do_dyn) {
      const float gx = s_goal[g * SD], gy = s_goal[g * SD + 1];
      float d2g;
      if (spread) {  // each goal finds the nearest agent (lidar_spread.py:41-44)
        d2g = 0.0f;
        for (int j = 0; j < n; ++j) {
          float dx = gx - s_agent[j * SD], dy = gy - s_agent[j * SD + 1];
          float d = sqrtf(dx * dx + dy * dy);
          d2g = (j == 0) ? d : nanmin(d2g, d);
        }
      } else {       // paired goal (lidar_target.py:41-44)
        float dx = gx - s_agent[g * SD], dy = gy - s_agent[g * SD + 1];
        d2g = sqrtf(dx * dx + dy * dy);
      }
      s_d2g[g] = d2g;
    }
  }
  __syncthreads();

  if (do_dyn && tid == 0) {
    float s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    for (int g = 0; g < ng; ++g) s1 = (g == 0) ? s_d2g[0] : s1 + s_d2g[g];
    for (int g = 0; g < ng; ++g) {
      float ind = (s_d2g[g] > c.dist2goal) ? 1.0f : 0.0f;
      s2 = (g == 0) ? ind : s2 + ind;
    }
    for (int i = 0; i < n; ++i) s3 = (i == 0) ? s_an2[0] : s3 + s_an2[i];
    float r = 0.0f;
    r = r - (s1 / (float)ng) * 0.01f;
    r = r - (s2 / (float)ng) * 0.001f;
    r = r - (s3 / (float)n) * 0.0001f;
    a.reward[b] = r;
  }

  // ---- phase 2: ray fan x rectangles x 4 segments (thread per (agent, ray)) ------------------
  if (do_sense) {
    const float sr = c.comm_radius;
    for (int idx = tid; idx < n * R; idx += nt) {
      const int i = idx / R, r = idx - i * R;
      const float x1 = s_next[i * SD], y1 = s_next[i * SD + 1];
      const float x2 = x1 + a.ray_cos[r] * sr;
      const float y2 = y1 + a.ray_sin[r] * sr;
      float amin = 0.0f;
      for (int o = 0; o < no; ++o) {
        const float* P = s_obst + o * DGPPO_RECT_STRIDE + 8;
        float ao = 0.0f;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int mm = (m + 3) & 3;
          const float x3 = P[2 * m], y3 = P[2 * m + 1], x4 = P[2 * mm], y4 = P[2 * mm + 1];
          float det = (x1 - x2) * (y4 - y3) - (y1 - y2) * (x4 - x3);
          float sgn = (det > 0.0f) ? 1.0f : ((det < 0.0f) ? -1.0f : det);
          det = sgn * fminf(fmaxf(fabsf(det), 1e-7f), 1e7f);
          float al = ((y4 - y3) * (x1 - x3) - (x4 - x3) * (y1 - y3)) / det;
          float be = ((-(y1 - y2)) * (x1 - x3) + (x1 - x2) * (y1 - y3)) / det;
          float v = ((al <= 1.0f) && (al >= 0.0f) && (be <= 1.0f) && (be >= 0.0f)) ? 1.0f : 0.0f;
          al = v * al + (1.0f - v) * 1e6f;
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
          mask = sqrtf(dx * dx + dy * dy) < c.comm_radius;
          sender = n + ng + m;
        }
      }
      edges[e] = f;
      recv[e] = mask ? i : pad;
      send[e] = mask ? sender : pad;
    }
  }
}

static size_t step_smem_bytes(const dgppo_env_cfg& c) {
  const int n = c.n_agents, ng = c.n_goals, no = c.n_obs, SD = c.state_dim;
  const bool lidar = cfg_is_lidar(c);
  const int kk = lidar ? (no > 0 ? c.top_k : 0) : 0;
  size_t fl = (size_t)n * SD * 2 + (size_t)ng * SD + n * 2 + (size_t)no * cfg_obst_stride(c) + (size_t)n * kk * 4 +
              (lidar ? (size_t)n * c.n_rays : 0) + n * 4 + ng * 4 + ng + n + n;
  return fl * sizeof(float);
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
  const size_t smem = step_smem_bytes(*cfg);
  DGPPO_REQUIRE(smem <= 64 * 1024, "env too large for the per-env LDS stage (%zu bytes)", smem);
  int work = cfg->n_agents * (lidar ? cfg->n_rays : 1);
  int E = cfg_num_edges(*cfg);
  if (gout && E > work) work = E;
  int threads = ((work + 63) / 64) * 64;
  if (threads < 64) threads = 64;
  if (threads > 512) threads = 512;
  hipStream_t s = (hipStream_t)stream;
  if (cfg->state_dim == 5)
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
