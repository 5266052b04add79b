// Gradient post-processing + optimiser over ONE flat fp32 buffer per network.
//
// Reference arithmetic replaced (file:line relative to /root/reference):
//   has_any_nan_or_inf, compute_norm_and_clip   dgppo/trainer/utils.py:89-118
//   optax.apply_if_finite(optax.adam(lr))        dgppo/algo/informarl.py:131-137,165-171 ; dgppo/algo/dgppo.py:99-105
//   TrainState.apply_gradients                   dgppo/algo/informarl.py:380,447 ; dgppo/algo/dgppo.py:319
// state[0] = sum g^2, state[1] = non-finite count, state[2] = adam count t (successful steps), state[3] = total steps,
// state[4] = last grad norm, state[5] = last non-finite flag  — all on the device, no host synchronisation.
// Every gradient entry is read as g * grad_scale (1/world after the all-reduce(sum) of the data-parallel path).
#include "common.h"

__device__ inline float wave_sum_f(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__global__ void __launch_bounds__(256) grad_stats_kernel(const float* __restrict__ g, long n, float* __restrict__ state,
                                                         float gscale) {
  float ss = 0.0f, bad = 0.0f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float v = g[i] * gscale;
    ss = fmaf(v, v, ss);
    bad += (isfinite(v) ? 0.0f : 1.0f);
  }
  ss = wave_sum_f(ss);
  bad = wave_sum_f(bad);
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(state + 0, ss);
    if (bad > 0.0f) atomicAdd(state + 1, bad);
  }
}

// g <- g / max(max_norm, ||g||) * max_norm ; Adam ; skipped entirely when any gradient entry is non-finite
__global__ void __launch_bounds__(256) clip_adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v, long n,
                                                        const float* __restrict__ state, float lr, float b1, float b2,
                                                        float eps, float max_norm, float gscale) {
  const float bad = state[1];
  if (bad > 0.0f) return;  // optax.apply_if_finite: zero update, inner state untouched
  const float norm = sqrtf(state[0]);
  const float denom = fmaxf(max_norm, norm);
  const float t = state[2] + 1.0f;
  const float c1 = 1.0f - powf(b1, t), c2 = 1.0f - powf(b2, t);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float gi = ((g[i] * gscale) / denom) * max_norm;
    const float mi = b1 * m[i] + (1.0f - b1) * gi;
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] = p[i] - lr * (mi / c1) / (sqrtf(vi / c2) + eps);
  }
}

__global__ void optim_finish_kernel(float* state) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const float bad = state[1];
    state[4] = sqrtf(state[0]);
    state[5] = (bad > 0.0f) ? 1.0f : 0.0f;
    if (!(bad > 0.0f)) state[2] += 1.0f;
    state[3] += 1.0f;
    state[0] = 0.0f;
    state[1] = 0.0f;
  }
}

extern "C" int32_t dgppo_clip_adam_step(float* params, const float* grads, float* m, float* v, int64_t n, float* state,
                                        float lr, float b1, float b2, float eps, float max_norm, float grad_scale,
                                        void* stream) {
  DGPPO_REQUIRE(n >= 0, "clip_adam: n < 0");
  if (n == 0) return 0;
  DGPPO_REQUIRE(params && grads && m && v && state, "clip_adam: NULL operand");
  hipStream_t s = (hipStream_t)stream;
  const int grid = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
  hipLaunchKernelGGL(grad_stats_kernel, dim3(grid), dim3(256), 0, s, grads, (long)n, state, grad_scale);
  hipLaunchKernelGGL(clip_adam_kernel, dim3(grid), dim3(256), 0, s, params, grads, m, v, (long)n, state, lr, b1, b2, eps,
                     max_norm, grad_scale);
  hipLaunchKernelGGL(optim_finish_kernel, dim3(1), dim3(64), 0, s, state);
  DGPPO_LAUNCH_CHECK();
  return 0;
}
