// Dec-OCP GAE and the advantage / CBF merge, one workgroup per environment.
//
// Reference arithmetic replaced (file:line relative to /root/reference):
//   compute_dec_ocp_gae        dgppo/algo/utils.py:11-79   (O(T^2) DP: max-discounted constraint rows + lambda weights)
//   advantage block            dgppo/algo/dgppo.py:239-259 (per-env normalisation, CBF derivative, safe gate, schedule)
// Layout (env-major): costs [B,T,n,nh], rewards [B,T], Vh [B,T+1,n,nh], Vl [B,T+1] -> Qh [B,T,n,nh], Ql [B,T].
#include "common.h"

struct GaeArgs {
  const float* costs; const float* rewards; const float* Vh; const float* Vl;
  const float* lam_pow;  // [T+1] lambda^i
  float* Qh; float* Ql;
  int B, T, AH;  // AH = n * nh
  int n, nh;
  float gamma, one_minus_gamma, one_minus_lam;
};

__global__ void gae_kernel(GaeArgs a) {
  extern __shared__ float sm[];
  const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const int T = a.T, AH = a.AH, n = a.n, nh = a.nh;
  float* rowH = sm;                       // [T+1][AH]
  float* rowL = rowH + (T + 1) * AH;      // [T+1]
  float* hcur = rowL + (T + 1);           // [AH]   hs[t]
  float* hdis = hcur + AH;                // [AH]   (1-gamma) * max_h hs[t]
  const float* costs = a.costs + (size_t)b * T * AH;
  const float* Vh = a.Vh + (size_t)b * (T + 1) * AH;
  const float* Vl = a.Vl + (size_t)b * (T + 1);
  const float* rew = a.rewards + (size_t)b * T;
  for (int i = tid; i < (T + 1) * AH; i += nt) rowH[i] = 0.0f;
  for (int i = tid; i < T + 1; i += nt) rowL[i] = 0.0f;
  __syncthreads();
  for (int i = tid; i < AH; i += nt) rowH[i] = Vh[(size_t)T * AH + i];   // row 0 <- V(x_T)   (utils.py:64-72)
  if (tid == 0) rowL[0] = Vl[T];
  __syncthreads();
  for (int ii = 0; ii < T; ++ii) {
    const int t = T - 1 - ii;
    for (int i = tid; i < AH; i += nt) hcur[i] = costs[(size_t)t * AH + i];
    __syncthreads();
    for (int ag = tid; ag < n; ag += nt) {  // discount towards max_h h  (utils.py:39-43)
      float m = hcur[ag * nh];
      for (int h = 1; h < nh; ++h) m = fmaxf(m, hcur[ag * nh + h]);
      for (int h = 0; h < nh; ++h) hdis[ag * nh + h] = a.one_minus_gamma * m;
    }
    __syncthreads();
    const float l = -rew[t];
    for (int idx = tid; idx < (ii + 1) * AH; idx += nt) {
      const int c = idx % AH;
      rowH[idx] = fmaxf(hcur[c], hdis[c] + a.gamma * rowH[idx]);
    }
    for (int j = tid; j <= ii; j += nt) rowL[j] = l + a.gamma * rowL[j];
    __syncthreads();
    // Q = sum_j c_j row_j,  c_0 = lambda^ii, c_j = lambda^(ii-j) (1 - lambda)   (utils.py:48-60)
    for (int c = tid; c < AH + 1; c += nt) {
      float acc = 0.0f;
      if (c < AH) {
        for (int j = 0; j <= ii; ++j) {
          const float cj = (j == 0) ? a.lam_pow[ii] : a.lam_pow[ii - j] * a.one_minus_lam;
          acc = fmaf(cj, rowH[j * AH + c], acc);
        }
        a.Qh[((size_t)b * T + t) * AH + c] = acc;
      } else {
        for (int j = 0; j <= ii; ++j) {
          const float cj = (j == 0) ? a.lam_pow[ii] : a.lam_pow[ii - j] * a.one_minus_lam;
          acc = fmaf(cj, rowL[j], acc);
        }
        a.Ql[(size_t)b * T + t] = acc;
      }
    }
    __syncthreads();
    for (int i = tid; i < AH; i += nt) rowH[(ii + 1) * AH + i] = Vh[(size_t)t * AH + i];   // utils.py:53-54
    if (tid == 0) rowL[ii + 1] = Vl[t];
    __syncthreads();
  }
}

extern "C" int32_t dgppo_gae(const float* costs, const float* rewards, const float* Vh, const float* Vl,
                             const float* lam_pow, float gamma, float one_minus_gamma, float one_minus_lam, float* Qh,
                             float* Ql, int32_t B, int32_t T, int32_t n, int32_t nh, void* stream) {
  DGPPO_REQUIRE(B >= 0 && T >= 1 && n >= 1 && nh >= 1, "gae: bad sizes");
  if (B == 0) return 0;
  DGPPO_REQUIRE(costs && rewards && Vh && Vl && lam_pow && Qh && Ql, "gae: NULL operand");
  GaeArgs a{costs, rewards, Vh, Vl, lam_pow, Qh, Ql, B, T, n * nh, n, nh, gamma, one_minus_gamma, one_minus_lam};
  const size_t smem = sizeof(float) * ((size_t)(T + 1) * a.AH + (T + 1) + 2 * a.AH);
  DGPPO_REQUIRE(smem <= 150 * 1024, "gae: T*n*nh too large for LDS (%zu B)", smem);
  if (smem > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)gae_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL(gae_kernel, dim3(B), dim3(256), smem, (hipStream_t)stream, a);
  DGPPO_LAUNCH_CHECK();
  return 0;
}

// ---- advantage merge (dgppo.py:239-259) -------------------------------------------------------------------------------
struct AdvArgs {
  const float* Ql; const float* Vl; const float* Vh;
  float* adv;      // [B,T,n]
  float* stats;    // stats[0] += number of safe (t, agent) pairs
  int B, T, n, nh;
  float inv_dt, alpha, cbf_eps, cbf_weight;
};

__global__ void adv_kernel(AdvArgs a) {
  __shared__ float red[256];
  __shared__ float s_mean, s_std;
  const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const int T = a.T, n = a.n, nh = a.nh;
  const float* Ql = a.Ql + (size_t)b * T;
  const float* Vl = a.Vl + (size_t)b * (T + 1);
  const float* Vh = a.Vh + (size_t)b * (T + 1) * n * nh;
  float acc = 0.0f;
  for (int t = tid; t < T; t += nt) acc += Ql[t] - Vl[t];
  red[tid] = acc;
  __syncthreads();
  for (int o = nt / 2; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
  if (tid == 0) s_mean = red[0] / (float)T;
  __syncthreads();
  const float mean = s_mean;
  acc = 0.0f;
  for (int t = tid; t < T; t += nt) { const float d = (Ql[t] - Vl[t]) - mean; acc += d * d; }
  red[tid] = acc;
  __syncthreads();
  for (int o = nt / 2; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
  if (tid == 0) s_std = sqrtf(red[0] / (float)T);
  __syncthreads();
  const float denom = s_std + 1e-8f;
  float nsafe = 0.0f;
  for (int idx = tid; idx < T * n; idx += nt) {
    const int t = idx / n, ag = idx - t * n;
    const float al = ((Ql[t] - Vl[t]) - mean) / denom;
    bool safe = true;
    float amax = 0.0f;
    for (int h = 0; h < nh; ++h) {
      const float v0 = Vh[((size_t)t * n + ag) * nh + h], v1 = Vh[((size_t)(t + 1) * n + ag) * nh + h];
      const float deriv = (v1 - v0) * a.inv_dt + a.alpha * v0;
      const float ac = fmaxf(deriv + a.cbf_eps, 0.0f);
      amax = (h == 0) ? ac : fmaxf(amax, ac);
      safe = safe && (deriv <= 0.0f);
    }
    const float A = (safe ? al : 0.0f) + amax * a.cbf_weight;
    a.adv[(size_t)b * T * n + idx] = -A;
    nsafe += safe ? 1.0f : 0.0f;
  }
  red[tid] = nsafe;
  __syncthreads();
  for (int o = nt / 2; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
  if (tid == 0) atomicAdd(a.stats, red[0]);
}

extern "C" int32_t dgppo_advantage(const float* Ql, const float* Vl, const float* Vh, float dt, float alpha,
                                   float cbf_eps, float cbf_weight, float* adv, float* stats, int32_t B, int32_t T,
                                   int32_t n, int32_t nh, void* stream) {
  DGPPO_REQUIRE(B >= 0 && T >= 1 && n >= 1 && nh >= 1, "advantage: bad sizes");
  if (B == 0) return 0;
  DGPPO_REQUIRE(Ql && Vl && Vh && adv && stats, "advantage: NULL operand");
  AdvArgs a{Ql, Vl, Vh, adv, stats, B, T, n, nh, 1.0f / dt, alpha, cbf_eps, cbf_weight};
  hipLaunchKernelGGL(adv_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, a);
  DGPPO_LAUNCH_CHECK();
  return 0;
}
