// Gradient post-processing + optimiser over ONE flat fp32 buffer per network.
//
// Reference arithmetic replaced (file:line relative to /root/reference):
//   has_any_nan_or_inf, compute_norm_and_clip   dgppo/trainer/utils.py:89-118
//   optax.apply_if_finite(optax.adam(lr))        dgppo/algo/informarl.py:131-137,165-171 ; dgppo/algo/dgppo.py:99-105
//   TrainState.apply_gradients                   dgppo/algo/informarl.py:380,447 ; dgppo/algo/dgppo.py:319
// state[DGPPO_OPT_STATE_FLOATS]: [2] = adam count t (successful steps), [3] = total steps, [4] = last grad norm,
// [5] = last non-finite flag, [8 .. 8 + 2*DGPPO_OPT_PARTIALS) = per-workgroup partials of the statistics pass — all on the
// device, no host synchronisation, no atomics (the norm is reduced in a fixed order).
// Every gradient entry is read as g * grad_scale (1/world after the all-reduce(sum) of the data-parallel path).
#include "common.h"

#define OPT_BLOCKS DGPPO_OPT_PARTIALS   // fixed grid of the statistics pass: one partial (sum g^2, #non-finite) per workgroup

__device__ inline float wave_sum_f(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);   // butterfly: every lane ends with the same bits
  return v;
}

// Sum of squares and non-finite count of g * gscale, in a FIXED order: grid-stride partial per thread, butterfly per
// wave, waves of a workgroup in index order, one partial per workgroup written to part[2*OPT_BLOCKS] (no atomics).
// The data-parallel replicas must stay bit-identical, so the norm may not depend on arrival order.
__global__ void __launch_bounds__(256) grad_stats_kernel(const float* __restrict__ g, long n, float* __restrict__ part,
                                                         float gscale) {
  __shared__ float s_ss[4], s_bad[4];
  float ss = 0.0f, bad = 0.0f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float v = g[i] * gscale;
    ss = fmaf(v, v, ss);
    bad += (isfinite(v) ? 0.0f : 1.0f);
  }
  ss = wave_sum_f(ss);
  bad = wave_sum_f(bad);
  if ((threadIdx.x & 63) == 0) { s_ss[threadIdx.x >> 6] = ss; s_bad[threadIdx.x >> 6] = bad; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[blockIdx.x] = ((s_ss[0] + s_ss[1]) + s_ss[2]) + s_ss[3];
    part[OPT_BLOCKS + blockIdx.x] = ((s_bad[0] + s_bad[1]) + s_bad[2]) + s_bad[3];
  }
}

// total of the OPT_BLOCKS partials in a fixed order (the same code in every workgroup of every kernel below, so every
// workgroup — and every rank — obtains the same bits): lane l adds partials l, l+64, l+128, ... then the butterfly.
__device__ inline void reduce_partials(const float* __restrict__ part, float& ss, float& bad) {
  const int l = threadIdx.x & 63;
  float a = 0.0f, b = 0.0f;
#pragma unroll
  for (int q = 0; q < OPT_BLOCKS / 64; ++q) { a += part[q * 64 + l]; b += part[OPT_BLOCKS + q * 64 + l]; }
  ss = wave_sum_f(a);
  bad = wave_sum_f(b);
}

// g <- g / max(max_norm, ||g||) * max_norm ; Adam ; skipped entirely when any gradient entry is non-finite
__global__ void __launch_bounds__(256) clip_adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v, long n,
                                                        const float* __restrict__ state, float lr, float b1, float b2,
                                                        float eps, float max_norm, float gscale) {
  float ss, bad;
  reduce_partials(state + 8, ss, bad);
  if (bad > 0.0f) return;  // optax.apply_if_finite: zero update, inner state untouched
  const float norm = sqrtf(ss);
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

__global__ void __launch_bounds__(64) optim_finish_kernel(float* state) {
  float ss, bad;
  reduce_partials(state + 8, ss, bad);
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    state[4] = sqrtf(ss);
    state[5] = (bad > 0.0f) ? 1.0f : 0.0f;
    if (!(bad > 0.0f)) state[2] += 1.0f;
    state[3] += 1.0f;
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
  // every one of the OPT_BLOCKS partial slots is (re)written each call, so the scratch half of `state` needs no zeroing
  hipLaunchKernelGGL(grad_stats_kernel, dim3(OPT_BLOCKS), dim3(256), 0, s, grads, (long)n, state + 8, grad_scale);
  hipLaunchKernelGGL(clip_adam_kernel, dim3(grid), dim3(256), 0, s, params, grads, m, v, (long)n, state, lr, b1, b2, eps,
                     max_norm, grad_scale);
  hipLaunchKernelGGL(optim_finish_kernel, dim3(1), dim3(64), 0, s, state);
  DGPPO_LAUNCH_CHECK();
  return 0;
}
