// Shared between env_step.hip (generic + workgroup-per-env LiDAR kernels) and env_wave.hip (wave-per-env LiDAR kernel).
// Both translation units are built with -ffp-contract=off (see Makefile): every fp32 operation is one IEEE operation.
#pragma once
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
  uint32_t rcp_n, rcp_k, rcp_no, rcp_no4;   // ceil(2^32 / d) for the index divisions of lidar_step_kernel (fdiv below)
  int B;
  float thr2_comm, thr2_lidar;              // sqrt_threshold(comm_radius), sqrt_threshold(lidar_mask_radius) (wave kernel)
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
// jnp.clip semantics: a NaN stays a NaN (fminf/fmaxf would return the bound).  Costs can be NaN when a hit point of the
// pre-step graph is NaN (ray parallel to an edge); states never are.
__device__ inline float clampf_nan(float x, float lo, float hi) { return (x != x) ? x : fminf(fmaxf(x, lo), hi); }

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


#define MISS_BITS 0x49742400u  // bits of 1e6f

// wave-per-env LiDAR kernel (env_wave.hip): returns true when it has an instantiation for this configuration and has
// enqueued the launch; false -> the caller falls back to the workgroup-per-env kernels
bool launch_lidar_wave(const StepArgs& a, hipStream_t s);
