// Batched reset (one thread per environment; executed once per rollout) and the N(0,1) noise generator.
//
// Reference arithmetic replaced (file:line relative to /root/reference):
//   LidarEnv.reset              dgppo/env/lidar_env/base.py:89-124
//   LidarBicycleTarget.reset    dgppo/env/lidar_env/lidar_bicycle_target.py:60-90
//   MPE.reset                   dgppo/env/mpe/base.py:81-127
//   get_node_goal_rng           dgppo/env/utils.py:139-244  (sequential rejection sampling, <=1024 tries, restart)
//   Rectangle.create / inside   dgppo/env/obstacle.py:39-72
// RNG: Philox-4x32-10, draw d of env with seed s = Philox(counter=(d,0,0,0), key=(lo32 s, hi32 s)), words 0,1
// -> two uniforms in [0,1) (same stream as oracle/env_np.py; JAX's threefry keys cannot be reproduced).
// Built with -ffp-contract=off like env_step.hip.
#include "common.h"

struct ResetArgs {
  dgppo_env_cfg cfg;
  const uint64_t* seeds;
  float* agent;
  float* goal;
  float* obst;
  int B;
  // task variants: thresholds formed on the host in double from the fp32 PARAMS and rounded once (oracle: reset_thresholds)
  float form_lo, form_hi;     // MPEFormation landmark box: comm_radius + 2 car_radius .. area - comm_radius - 2 car_radius
  float line_obs_margin;      // LidarLine: 1.1 car_radius
  int* n_failed;              // optional: += 1 per env whose bounded rejection loops ran out (the scene is then invalid)
};

struct Stream {
  uint32_t k0, k1, d;
  __device__ inline void uniform2(float& u0, float& u1) {
    Philox4 p = philox4x32_10(d, 0u, 0u, 0u, k0, k1);
    d += 1;
    u0 = u01_from_u32(p.v[0]);
    u1 = u01_from_u32(p.v[1]);
  }
};

__device__ inline bool rect_inside_r(const float* rec, float px, float py, float r) {
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

// One rejection-sampled node (get_node / non_valid_node, env/utils.py:160-172): redraw until it is farther than
// min_dist from every row of `all` (unfilled rows are zeros) and outside every inflated rectangle, or max_iter tries.
__device__ __noinline__ void sample_node(Stream& st, float* all, int n, int SD, float A, float Ay, float min_dist, float half,
                                         const float* rects, int no, int max_iter, int slot, int& n_iter) {
  int it = 0;
  float cx = 0.0f, cy = 0.0f;
  while (true) {
    float u0, u1;
    st.uniform2(u0, u1);
    cx = u0 * A;
    cy = u1 * Ay;
    float dmin = 3.4e38f;
    for (int j = 0; j < n; ++j) {
      float dx = all[j * SD] - cx, dy = all[j * SD + 1] - cy;
      dmin = fminf(dmin, sqrtf(dx * dx + dy * dy));
    }
    bool inside = false;
    if (rects != nullptr)
      for (int o = 0; o < no; ++o) inside = inside || rect_inside_r(rects + o * DGPPO_RECT_STRIDE, cx, cy, half);
    const bool ok = !(dmin <= min_dist) && !inside;
    if (ok || it >= max_iter) break;
    it += 1;
  }
  all[slot * SD] = cx;
  all[slot * SD + 1] = cy;
  n_iter = it;
}

#define MAX_AGENTS 64

// ---- task variants (lidar_line.py:39-129, mpe_line.py:36-122, mpe_formation.py:37-92, mpe_corridor.py:41-60,
//      mpe_connect_spread.py:50-107): same stream, statement order of oracle/env_np.py::_reset_variant -------------------
// positions the reward measures against (the same arithmetic as env_step.hip phase 1a / oracle reward_goal_positions)
__device__ inline void reward_goal(const dgppo_env_cfg& c, const float* lm, int SD, int g, float& gx, float& gy) {
  const int n = c.n_agents;
  if (c.reward_goals == DGPPO_GOALS_CIRCLE) {
    const float th = ((float)g / (float)n) * 6.28318530717958647692f;
    gx = lm[0] + c.comm_radius * cosf(th);
    gy = lm[1] + c.comm_radius * sinf(th);
  } else {
    const bool ends = c.reward_goals == DGPPO_GOALS_LINE;
    const float i_f = ends ? (float)g : (float)(g + 1), den = ends ? (float)(n - 1) : (float)(n + 1);
    gx = lm[0] + (i_f * (lm[SD] - lm[0])) / den;
    gy = lm[1] + (i_f * (lm[SD + 1] - lm[1])) / den;
  }
}

// nearest-neighbour distance of row i among the n rows of p (+1e6 on the diagonal, mpe_connect_spread.py:54-56)
__device__ inline float nn_dist(const float* p, int n, int SD, int i) {
  float m = 3.4e38f;
  for (int j = 0; j < n; ++j) {
    const float dx = p[i * SD] - p[j * SD], dy = p[i * SD + 1] - p[j * SD + 1];
    m = fminf(m, sqrtf(dx * dx + dy * dy) + ((i == j) ? 1e6f : 0.0f));
  }
  return m;
}

__global__ void env_reset_variant_kernel(ResetArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.B) return;
  const dgppo_env_cfg& c = a.cfg;
  const int n = c.n_agents, ng = c.n_goals, no = c.n_obs, SD = c.state_dim;
  const bool lidar = cfg_is_lidar(c);
  const float A = c.area_size;
  const uint64_t seed = a.seeds[b];
  Stream st;
  st.k0 = (uint32_t)seed; st.k1 = (uint32_t)(seed >> 32); st.d = 0;
  float* agent = a.agent + (size_t)b * n * SD;
  float* goal = a.goal + (size_t)b * ng * SD;
  float* obst = a.obst ? a.obst + (size_t)b * no * cfg_obst_stride(c) : nullptr;
  // work array for the goals get_node_goal_rng samples next to the agents: the kinds with n goal nodes use the output
  // rows; Line / Formation discard those goals (states, _ = get_node_goal_rng(...)): positions only, in LDS
  __shared__ float s_discard[64][MAX_AGENTS * 2];
  const bool own_goals = (ng == n);
  float* sgoal = own_goals ? goal : s_discard[threadIdx.x];
  const int GS = own_goals ? SD : 2;                  // row stride of sgoal
  const float min_dist = c.reset_min_dist, half = min_dist / 2.0f;
  const int max_iter = 1024;
  const bool connect = c.kind == DGPPO_ENV_MPE_CONNECT_SPREAD;
  // The reference's rejection loops are unbounded; here every loop has a bound so that every thread reaches the exit.  A
  // scene whose bound ran out is INVALID (colliding / disconnected agents): it is counted in *n_failed, which the caller
  // checks at its next host sync (dgppo_env_reset_checked) — never silently used.
  bool gave_up = false;
  // ---- agents (and sampled goals): bounded restarts, and for ConnectSpread the bounded connectivity rejection ----
  bool placed = false, connected = !connect;
  for (int outer = 0; outer < (connect ? 4096 : 1); ++outer) {
    placed = false;
    for (int attempt = 0; attempt < 64; ++attempt) {
      for (int i = 0; i < n * SD; ++i) agent[i] = 0.0f;
      for (int i = 0; i < n * GS; ++i) sgoal[i] = 0.0f;
      bool failed = false;
      for (int i = 0; i < n; ++i) {
        int it_a = 0, it_g = 0;
        sample_node(st, agent, n, SD, A, c.reset_side_y, min_dist, half, nullptr, 0, max_iter, i, it_a);
        sample_node(st, sgoal, n, GS, A, c.reset_side_y, min_dist, half, nullptr, 0, max_iter, i, it_g);
        if (it_a >= max_iter || it_g >= max_iter) { failed = true; break; }
      }
      if (!failed) { placed = true; break; }
    }
    if (!connect || !placed) break;                  // 64 restarts without a valid placement: give up (infeasible density)
    bool bad = false;
    for (int i = 0; i < n; ++i) {
      const float mda = nn_dist(agent, n, SD, i), mdg = nn_dist(sgoal, n, GS, i);
      bad = bad || (mda > c.connect_radius) || (mda < c.two_car_radius) || (mdg > c.connect_radius);
    }
    if (!bad) { connected = true; break; }
  }
  gave_up = !placed || !connected;
  if (c.kind == DGPPO_ENV_MPE_CORRIDOR || connect) {
    for (int i = 0; i < n; ++i) goal[i * SD + 1] = goal[i * SD + 1] + c.goal_shift_y;
    if (connect) {                                   // one large disc, x uniform (mpe_connect_spread.py:91-95)
      float u0, u1;
      st.uniform2(u0, u1);
      for (int d = 0; d < SD; ++d) obst[d] = 0.0f;
      obst[0] = c.obs_radius + u0 * ((A - c.obs_radius) - c.obs_radius);
      obst[1] = A / 2.0f;
    } else {                                         // the two corridor walls (mpe_corridor.py:55-56)
      for (int d = 0; d < 2 * SD; ++d) obst[d] = 0.0f;
      obst[0] = c.obs_radius; obst[1] = A / 2.0f;
      obst[SD] = A - c.obs_radius; obst[SD + 1] = A / 2.0f;
    }
    if (gave_up && a.n_failed) atomicAdd(a.n_failed, 1);
    return;
  }
  // ---- landmarks ----
  for (int i = 0; i < ng * SD; ++i) goal[i] = 0.0f;
  if (c.kind == DGPPO_ENV_MPE_FORMATION) {
    const float lo = a.form_lo, hi = a.form_hi;
    float u0, u1;
    st.uniform2(u0, u1);
    goal[0] = lo + u0 * (hi - lo);
    goal[1] = lo + u1 * (hi - lo);
  } else {
    const float md = c.line_min_dist;
    float u0, u1;
    if (c.kind == DGPPO_ENV_MPE_LINE && n <= 3) {
      st.uniform2(u0, u1);
      goal[0] = u0 * A; goal[1] = u1 * A;
    } else {
      const float side = A - md;
      st.uniform2(u0, u1);
      const float cx = u0 * (A - side) - A / 2.0f;
      const float cy = u1 * side + (A / 2.0f - side);
      st.uniform2(u0, u1);
      int region = (int)(u0 * 4.0f);
      region = region > 3 ? 3 : region;
      float rx, ry;                                  // exact quarter turns (see oracle/_reset_variant)
      if (region == 0) { rx = cx; ry = cy; } else if (region == 1) { rx = -cy; ry = cx; }
      else if (region == 2) { rx = -cx; ry = -cy; } else { rx = cy; ry = -cx; }
      goal[0] = rx + A / 2.0f; goal[1] = ry + A / 2.0f;
    }
    bool found = false;
    for (int it = 0; it < 100000 && !found; ++it) {
      st.uniform2(u0, u1);
      goal[SD] = u0 * A; goal[SD + 1] = u1 * A;
      const float dx = goal[SD] - goal[0], dy = goal[SD + 1] - goal[1];
      found = !(sqrtf(dx * dx + dy * dy) < md);
    }
    gave_up = gave_up || !found;
  }
  // ---- obstacles: keep clear of the agents and of the n reward goals ----
  if (lidar) {
    const float r_in = a.line_obs_margin;
    for (int o = 0; o < no; ++o) {
      float* rec = obst + o * DGPPO_RECT_STRIDE;
      bool found = false;
      for (int it = 0; it < 100000; ++it) {
        float u0, u1;
        st.uniform2(u0, u1);
        const float cx = u0 * A, cy = u1 * A;
        st.uniform2(u0, u1);
        const float lo = 0.1f, hi = 0.3f;
        const float w = lo + u0 * (hi - lo), h = lo + u1 * (hi - lo);
        st.uniform2(u0, u1);
        const float th = u0 * 3.14159265358979323846f;
        const float cs = cosf(th), sn = sinf(th);
        rec[0] = cx; rec[1] = cy; rec[2] = w; rec[3] = h; rec[4] = th; rec[5] = cs; rec[6] = sn; rec[7] = 0.0f;
        const float hw = w / 2.0f, hh = h / 2.0f;
        const float bx[4] = {hw, -hw, -hw, hw};
        const float by[4] = {hh, hh, -hh, -hh};
        for (int m = 0; m < 4; ++m) {
          rec[8 + 2 * m] = (cs * bx[m] + (-sn) * by[m]) + cx;
          rec[9 + 2 * m] = (sn * bx[m] + cs * by[m]) + cy;
        }
        bool inside = false;
        for (int i = 0; i < n; ++i) inside = inside || rect_inside_r(rec, agent[i * SD], agent[i * SD + 1], r_in);
        for (int g = 0; g < n; ++g) {
          float gx, gy;
          reward_goal(c, goal, SD, g, gx, gy);
          inside = inside || rect_inside_r(rec, gx, gy, r_in);
        }
        if (!inside) { found = true; break; }
      }
      gave_up = gave_up || !found;
    }
  } else {
    const float lo = c.car_radius * 3.0f;
    const float hi = A - c.car_radius * 3.0f;
    const float thr_g = c.two_car_radius + c.obs_radius;
    for (int o = 0; o < no; ++o) {
      bool first = true, found = false;
      float cx = 0.0f, cy = 0.0f;
      for (int it = 0; it < 100000; ++it) {
        float u0, u1;
        st.uniform2(u0, u1);
        if (first) { cx = u0 * A; cy = u1 * A; first = false; }
        else { cx = lo + u0 * (hi - lo); cy = lo + u1 * (hi - lo); }
        float da = 3.4e38f, dg = 3.4e38f;
        for (int j = 0; j < n; ++j) {
          float dx = agent[j * SD] - cx, dy = agent[j * SD + 1] - cy;
          da = fminf(da, sqrtf(dx * dx + dy * dy));
          float gx, gy;
          reward_goal(c, goal, SD, j, gx, gy);
          dx = gx - cx; dy = gy - cy;
          dg = fminf(dg, sqrtf(dx * dx + dy * dy));
        }
        const bool bad = (da <= c.car_plus_obs) || (dg <= thr_g) || (cx < lo) || (cy < lo) || (cx > hi) || (cy > hi);
        if (!bad) { found = true; break; }
      }
      gave_up = gave_up || !found;
      for (int d = 0; d < SD; ++d) obst[o * SD + d] = 0.0f;
      obst[o * SD] = cx; obst[o * SD + 1] = cy;
    }
  }
  if (gave_up && a.n_failed) atomicAdd(a.n_failed, 1);
}


__global__ void env_reset_kernel(ResetArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.B) return;
  const dgppo_env_cfg& c = a.cfg;
  const int n = c.n_agents, no = c.n_obs, SD = c.state_dim;
  const bool lidar = cfg_is_lidar(c), bicycle = cfg_is_bicycle(c);
  const float A = c.area_size;
  const uint64_t seed = a.seeds[b];
  Stream st;
  st.k0 = (uint32_t)seed;
  st.k1 = (uint32_t)(seed >> 32);
  st.d = 0;
  float* agent = a.agent + (size_t)b * n * SD;
  float* goal = a.goal + (size_t)b * n * SD;
  float* obst = a.obst ? a.obst + (size_t)b * no * cfg_obst_stride(c) : nullptr;
  const float PI_F = 3.14159265358979323846f;
  const float TWO_PI_F = 6.28318530717958647692f;

  if (lidar) {
    for (int o = 0; o < no; ++o) {  // lidar_env/base.py:96-108
      float u0, u1;
      st.uniform2(u0, u1);
      float cx = u0 * A, cy = u1 * A;
      st.uniform2(u0, u1);
      const float lo = 0.1f, hi = 0.3f;
      float w = lo + u0 * (hi - lo);
      float h = lo + u1 * (hi - lo);
      st.uniform2(u0, u1);
      float th = u0 * TWO_PI_F;
      if (bicycle) th = th - PI_F;
      float cs = cosf(th), sn = sinf(th);
      float* rec = obst + o * DGPPO_RECT_STRIDE;
      rec[0] = cx; rec[1] = cy; rec[2] = w; rec[3] = h; rec[4] = th; rec[5] = cs; rec[6] = sn; rec[7] = 0.0f;
      float hw = w / 2.0f, hh = h / 2.0f;
      float bx[4] = {hw, -hw, -hw, hw};
      float by[4] = {hh, hh, -hh, -hh};
      for (int m = 0; m < 4; ++m) {  // obstacle.py:40-54
        rec[8 + 2 * m] = (cs * bx[m] + (-sn) * by[m]) + cx;
        rec[9 + 2 * m] = (sn * bx[m] + cs * by[m]) + cy;
      }
    }
  }
  const float min_dist = c.reset_min_dist;
  const float half = min_dist / 2.0f;
  const int max_iter = 1024;
  // The output rows double as the working arrays all_states / all_goals of get_node_goal_rng
  // (env/utils.py:150-151): zero rows included, so the origin's neighbourhood is excluded as in the reference.
  // bounded restarts: every thread reaches the exit; a scene that fails 64 times keeps its last draw and is counted in
  // *n_failed (see env_reset_variant_kernel)
  bool gave_up = true;
  for (int attempt = 0; attempt < 64; ++attempt) {
    for (int i = 0; i < n * SD; ++i) { agent[i] = 0.0f; goal[i] = 0.0f; }
    bool failed = false;
    for (int i = 0; i < n; ++i) {
      int it_a = 0, it_g = 0;
      sample_node(st, agent, n, SD, A, A, min_dist, half, lidar ? obst : nullptr, no, max_iter, i, it_a);
      sample_node(st, goal, n, SD, A, A, min_dist, half, lidar ? obst : nullptr, no, max_iter, i, it_g);
      if (it_a >= max_iter || it_g >= max_iter) { failed = true; break; }
    }
    if (!failed) { gave_up = false; break; }
  }
  if (bicycle) {  // lidar_bicycle_target.py:80-83
    for (int i = 0; i < n; ++i) {
      float u0, u1;
      st.uniform2(u0, u1);
      float th = u0 * TWO_PI_F;
      agent[i * SD + 2] = cosf(th);
      agent[i * SD + 3] = sinf(th);
    }
  }
  if (!lidar) {  // mpe/base.py:93-118
    const float lo = c.car_radius * 3.0f;
    const float hi = A - c.car_radius * 3.0f;
    const float thr_g = c.two_car_radius + c.obs_radius;
    for (int o = 0; o < no; ++o) {
      bool first = true, found = false;
      float cx = 0.0f, cy = 0.0f;
      for (int it = 0; it < 100000; ++it) {
        float u0, u1;
        st.uniform2(u0, u1);
        if (first) { cx = u0 * A; cy = u1 * A; first = false; }
        else { cx = lo + u0 * (hi - lo); cy = lo + u1 * (hi - lo); }
        float da = 3.4e38f, dg = 3.4e38f;
        for (int j = 0; j < n; ++j) {
          float dx = agent[j * SD] - cx, dy = agent[j * SD + 1] - cy;
          da = fminf(da, sqrtf(dx * dx + dy * dy));
          dx = goal[j * SD] - cx; dy = goal[j * SD + 1] - cy;
          dg = fminf(dg, sqrtf(dx * dx + dy * dy));
        }
        bool bad = (da <= c.car_plus_obs) || (dg <= thr_g) || (cx < lo) || (cy < lo) || (cx > hi) || (cy > hi);
        if (!bad) { found = true; break; }
      }
      gave_up = gave_up || !found;
      for (int d = 0; d < SD; ++d) obst[o * SD + d] = 0.0f;
      obst[o * SD] = cx; obst[o * SD + 1] = cy;
    }
  }
  if (gave_up && a.n_failed) atomicAdd(a.n_failed, 1);
}

extern "C" int32_t dgppo_env_reset(const dgppo_env_cfg* cfg, const uint64_t* seeds, float* agent, float* goal, float* obst,
                                   int32_t B, void* stream) {
  return dgppo_env_reset_checked(cfg, seeds, agent, goal, obst, nullptr, B, stream);
}

extern "C" int32_t dgppo_env_reset_checked(const dgppo_env_cfg* cfg, const uint64_t* seeds, float* agent, float* goal,
                                           float* obst, int32_t* n_failed, int32_t B, void* stream) {
  int32_t rc = dgppo_validate_cfg(cfg);
  if (rc) return rc;
  {
    // feasibility of the rejection sampling (env/utils.py:139-244): n points with pairwise distance >= reset_min_dist in
    // [0, area] x [0, reset_side_y].  Random sequential placement of discs jams at ~0.547 coverage; beyond one half the loops
    // would run to their bounds on (nearly) every env — a launch that looks like a hang and yields invalid scenes.
    const double d = cfg->reset_min_dist, ax = cfg->area_size, ay = cfg->reset_side_y > 0.0f ? cfg->reset_side_y : cfg->area_size;
    const double cover = cfg->n_agents * 3.14159265358979 * (d / 2) * (d / 2), room = (ax + d) * (ay + d);
    DGPPO_REQUIRE(cover <= 0.5 * room,
                  "env_reset: %d agents with minimum separation %.3f cannot be placed in %.2f x %.2f by rejection sampling "
                  "(disc coverage %.2f of the area; the limit used here is 0.50)", cfg->n_agents, d, ax, ay, cover / room);
  }
  DGPPO_REQUIRE(B >= 0, "B must be >= 0");
  if (B == 0) return 0;
  DGPPO_REQUIRE(seeds && agent && goal, "seeds/agent/goal must not be NULL");
  DGPPO_REQUIRE(cfg->n_obs == 0 || obst, "obst must not be NULL when n_obs > 0");
  DGPPO_REQUIRE(cfg->n_agents <= MAX_AGENTS, "reset supports at most %d agents", MAX_AGENTS);
  ResetArgs a;
  a.cfg = *cfg; a.seeds = seeds; a.agent = agent; a.goal = goal; a.obst = obst; a.B = B; a.n_failed = n_failed;
  a.form_lo = (float)((double)cfg->comm_radius + 2.0 * (double)cfg->car_radius);
  a.form_hi = (float)((double)cfg->area_size - (double)cfg->comm_radius - 2.0 * (double)cfg->car_radius);
  a.line_obs_margin = (float)((double)cfg->car_radius * 1.1);
  if (cfg->kind >= DGPPO_ENV_LIDAR_LINE) {
    DGPPO_REQUIRE(!(cfg->kind == DGPPO_ENV_LIDAR_LINE || (cfg->kind == DGPPO_ENV_MPE_LINE && cfg->n_agents > 3)) ||
                      cfg->area_size - cfg->line_min_dist >= 0.0f,
                  "The area size is too small to place the landmarks.");      // lidar_line.py:56-57
    hipLaunchKernelGGL(env_reset_variant_kernel, dim3(cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, a);
  } else
    hipLaunchKernelGGL(env_reset_kernel, dim3(cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, a);
  DGPPO_LAUNCH_CHECK();
  return 0;
}

// ---- N(0,1) noise: Philox + Box-Muller; element i uses counter (lo32(off+i/4), hi32(off+i/4), 0, 1) ----------------
__global__ void randn_kernel(uint64_t seed, uint64_t offset, float* out, int64_t n_elem) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one Philox block -> 4 normals
  if (q * 4 >= n_elem) return;
  const uint64_t ctr = offset + (uint64_t)q;
  Philox4 p = philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 1u, (uint32_t)seed, (uint32_t)(seed >> 32));
  float z[4];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    // u1 in (0,1], u2 in [0,1)
    float u1 = ((float)(p.v[2 * h] >> 8) + 1.0f) * (1.0f / 16777216.0f);
    float u2 = u01_from_u32(p.v[2 * h + 1]);
    float r = sqrtf(-2.0f * logf(u1));
    float s, cth;
    sincosf(6.28318530717958647692f * u2, &s, &cth);
    z[2 * h] = r * cth;
    z[2 * h + 1] = r * s;
  }
  for (int j = 0; j < 4; ++j)
    if (q * 4 + j < n_elem) out[q * 4 + j] = z[j];
}

// one thread per (row, Philox block of the GLOBAL element index that overlaps the row's window)
__global__ void randn_rows_kernel(uint64_t seed, float* out, int64_t rows, int64_t row_len, int64_t global_row_len,
                                  int64_t col_offset, int64_t quads_per_row) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * quads_per_row) return;
  const int64_t r = i / quads_per_row, j = i - r * quads_per_row;
  const int64_t g0 = r * global_row_len + col_offset;           // first global element of this row's window
  const uint64_t ctr = (uint64_t)(g0 / 4 + j);
  Philox4 p = philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 1u, (uint32_t)seed, (uint32_t)(seed >> 32));
  float z[4];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    float u1 = ((float)(p.v[2 * h] >> 8) + 1.0f) * (1.0f / 16777216.0f);
    float u2 = u01_from_u32(p.v[2 * h + 1]);
    float rr = sqrtf(-2.0f * logf(u1));
    float s, cth;
    sincosf(6.28318530717958647692f * u2, &s, &cth);
    z[2 * h] = rr * cth;
    z[2 * h + 1] = rr * s;
  }
  for (int k = 0; k < 4; ++k) {
    const int64_t c = (int64_t)ctr * 4 + k - g0;                 // column inside the window
    if (c >= 0 && c < row_len) out[r * row_len + c] = z[k];
  }
}

extern "C" int32_t dgppo_randn_rows(uint64_t seed, float* out, int64_t rows, int64_t row_len, int64_t global_row_len,
                                    int64_t col_offset, void* stream) {
  DGPPO_REQUIRE(rows >= 0 && row_len >= 0 && col_offset >= 0 && col_offset + row_len <= global_row_len,
                "randn_rows: the window [col_offset, col_offset + row_len) must lie inside a row of global_row_len");
  if (rows == 0 || row_len == 0) return 0;
  DGPPO_REQUIRE(out, "out must not be NULL");
  const int64_t quads_per_row = row_len / 4 + 2;                  // covers any alignment of the window
  DGPPO_REQUIRE(rows * quads_per_row < ((int64_t)1 << 38), "randn_rows: too many elements");
  hipLaunchKernelGGL(randn_rows_kernel, dim3((unsigned)cdiv(rows * quads_per_row, 256)), dim3(256), 0, (hipStream_t)stream,
                     seed, out, rows, row_len, global_row_len, col_offset, quads_per_row);
  DGPPO_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t dgppo_randn(uint64_t seed, uint64_t offset, float* out, int64_t n_elem, void* stream) {
  DGPPO_REQUIRE(n_elem >= 0, "n_elem must be >= 0");
  if (n_elem == 0) return 0;
  DGPPO_REQUIRE(out, "out must not be NULL");
  const int64_t quads = (n_elem + 3) / 4;
  hipLaunchKernelGGL(randn_kernel, dim3(cdiv(quads, 256)), dim3(256), 0, (hipStream_t)stream, seed, offset, out, n_elem);
  DGPPO_LAUNCH_CHECK();
  return 0;
}
