// Shared host/device helpers for libdgppo_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "../../include/dgppo_hip.h"

#define DGPPO_WAVE 64

// ---- error plumbing (host) -------------------------------------------------------------------
void dgppo_set_error(const char* fmt, ...);

#define DGPPO_REQUIRE(cond, ...)            \
  do {                                      \
    if (!(cond)) {                          \
      dgppo_set_error(__VA_ARGS__);         \
      return -1;                            \
    }                                       \
  } while (0)
#define DGPPO_LAUNCH_CHECK()                                                   \
  do {                                                                         \
    hipError_t e__ = hipGetLastError();                                        \
    if (e__ != hipSuccess) {                                                   \
      dgppo_set_error("%s:%d launch failed: %s", __FILE__, __LINE__,           \
                      hipGetErrorString(e__));                                 \
      return (int32_t)e__;                                                     \
    }                                                                          \
  } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- env config helpers (host + device) ------------------------------------------------------
__host__ __device__ inline bool cfg_is_lidar(const dgppo_env_cfg& c) {
  return c.kind <= DGPPO_ENV_LIDAR_BICYCLE_TARGET || c.kind == DGPPO_ENV_LIDAR_LINE;
}
// every agent is connected to every goal node (Spread and its subclasses) vs to its own goal (Target)
__host__ __device__ inline bool cfg_is_spread(const dgppo_env_cfg& c) {
  return c.kind != DGPPO_ENV_LIDAR_TARGET && c.kind != DGPPO_ENV_LIDAR_BICYCLE_TARGET && c.kind != DGPPO_ENV_MPE_TARGET;
}
// the five base kinds with their own goal nodes as reward goals: what the specialised LiDAR kernels assume
__host__ __device__ inline bool cfg_is_base_kind(const dgppo_env_cfg& c) {
  return c.kind <= DGPPO_ENV_MPE_TARGET && c.n_goals == c.n_agents && c.reward_goals == DGPPO_GOALS_NODES && c.n_cost == 2;
}
// positions the reward measures against: n derived goals, or the goal nodes
__host__ __device__ inline int cfg_reward_goals(const dgppo_env_cfg& c) {
  return c.reward_goals == DGPPO_GOALS_NODES ? c.n_goals : c.n_agents;
}
__host__ __device__ inline bool cfg_is_bicycle(const dgppo_env_cfg& c) { return c.kind == DGPPO_ENV_LIDAR_BICYCLE_TARGET; }
// nodes that carry obstacle information: n*k LiDAR hit nodes or n_obs disc nodes
__host__ __device__ inline int cfg_obs_nodes(const dgppo_env_cfg& c) {
  if (cfg_is_lidar(c)) return c.n_obs > 0 ? c.n_agents * c.top_k : 0;
  return c.n_obs;
}
// obstacle slots seen by ONE agent: its own k hits (LiDAR) or all n_obs discs (MPE)
__host__ __device__ inline int cfg_obs_slots(const dgppo_env_cfg& c) {
  if (cfg_is_lidar(c)) return c.n_obs > 0 ? c.top_k : 0;
  return c.n_obs;
}
__host__ __device__ inline int cfg_goal_slots(const dgppo_env_cfg& c) { return cfg_is_spread(c) ? c.n_goals : 1; }
__host__ __device__ inline int cfg_num_nodes(const dgppo_env_cfg& c) { return c.n_agents + c.n_goals + cfg_obs_nodes(c) + 1; }
__host__ __device__ inline int cfg_num_edges(const dgppo_env_cfg& c) {
  return c.n_agents * (c.n_agents + cfg_goal_slots(c) + cfg_obs_slots(c));
}
__host__ __device__ inline int cfg_obst_stride(const dgppo_env_cfg& c) { return cfg_is_lidar(c) ? DGPPO_RECT_STRIDE : c.state_dim; }

int32_t dgppo_validate_cfg(const dgppo_env_cfg* cfg);

// ---- Philox-4x32-10 (stream layout documented in oracle/env_np.py) ---------------------------
struct Philox4 {
  uint32_t v[4];
};
__host__ __device__ inline Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)M0 * c0;
    uint64_t p1 = (uint64_t)M1 * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0; k1 += W1;
  }
  Philox4 o;
  o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
  return o;
}
__host__ __device__ inline float u01_from_u32(uint32_t w) { return (float)(w >> 8) * (1.0f / 16777216.0f); }

// NaN-propagating min (jnp.min semantics)
__device__ inline float nanmin(float a, float b) { return (a != a || b != b) ? __builtin_nanf("") : fminf(a, b); }
