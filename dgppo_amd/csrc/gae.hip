// Dec-OCP GAE and the advantage / CBF merge, one workgroup per environment.
//
// Reference arithmetic replaced (file:line relative to /root/reference):
//   compute_dec_ocp_gae        dgppo/algo/utils.py:11-79   (O(T^2) DP: max-discounted constraint rows + lambda weights)
//   advantage block            dgppo/algo/dgppo.py:239-259 (per-env normalisation, CBF derivative, safe gate, schedule)
// Layout (env-major): costs [B,T,n,nh], rewards [B,T], Vh [B,T+1,n,nh], Vl [B,T+1] -> Qh [B,T,n,nh], Ql [B,T].
#include "common.h"
#include <stdlib.h>

// jnp.maximum / .max(-1) propagate NaN (algo/utils.py:39-44); v_max_f32 (fmaxf) returns the other operand.  Costs can be NaN
// (a NaN LiDAR hit point, lidar_env/base.py:180-207), and the reference then gets NaN targets, NaN losses and a skipped
// optimiser step (optax.apply_if_finite) — so must this build.
__device__ inline float nanmax(float a, float b) { return (a != a || b != b) ? __builtin_nanf("") : fmaxf(a, b); }

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
      for (int h = 1; h < nh; ++h) m = nanmax(m, hcur[ag * nh + h]);
      for (int h = 0; h < nh; ++h) hdis[ag * nh + h] = a.one_minus_gamma * m;
    }
    __syncthreads();
    const float l = -rew[t];
    for (int idx = tid; idx < (ii + 1) * AH; idx += nt) {
      const int c = idx % AH;
      rowH[idx] = nanmax(hcur[c], hdis[c] + a.gamma * rowH[idx]);
    }
    // a NaN cost (or reward) turns every ACTIVE row of its column NaN; row 0 is always active and stays in every later sum,
    // so Q of this and of all earlier time steps is NaN — what `mask * maximum(...)` (utils.py:44, 0 * NaN = NaN) yields too
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


// Faster variant for T + 1 <= 256: thread j owns DP row j (its n*nh constraint values + the cost value) in registers,
// the env's costs / values are staged in LDS once, and Q = sum_j c_j row_j is a workgroup reduction (wave shuffles, then
// one LDS hop) instead of a serial loop.  Same recurrences; only the summation order of Q differs (fp32, ~1e-7).
__device__ inline float gae_row16_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, false));   // row_half_mirror
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, false));   // row_mirror
  return v;
}

template <int AHP>
__global__ void __launch_bounds__(256) gae_rows_kernel(GaeArgs a) {
  extern __shared__ float sm[];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int T = a.T, AH = a.AH, nh = a.nh;
  float* s_cost = sm;                       // [T][AH]
  float* s_Vh = s_cost + T * AH;            // [T+1][AH]
  float* s_Vl = s_Vh + (T + 1) * AH;        // [T+1]
  float* s_rew = s_Vl + (T + 1);            // [T]
  float* s_part = s_rew + T;                // [2][16][AHP+1]  per-16-lane-row partial sums, double buffered
  const float* costs = a.costs + (size_t)b * T * AH;
  const float* Vh = a.Vh + (size_t)b * (T + 1) * AH;
  for (int i = tid; i < T * AH; i += 256) s_cost[i] = costs[i];
  for (int i = tid; i < (T + 1) * AH; i += 256) s_Vh[i] = Vh[i];
  for (int i = tid; i < T + 1; i += 256) s_Vl[i] = a.Vl[(size_t)b * (T + 1) + i];
  for (int i = tid; i < T; i += 256) s_rew[i] = a.rewards[(size_t)b * T + i];
  __syncthreads();
  const int j = tid;                        // DP row of this thread
  float row[AHP];
  float rowl = 0.0f;
#pragma unroll
  for (int c = 0; c < AHP; ++c) row[c] = (j == 0 && c < AH) ? s_Vh[T * AH + c] : 0.0f;   // row 0 <- V(x_T)
  if (j == 0) rowl = s_Vl[T];
  for (int ii = 0; ii < T; ++ii) {
    const int t = T - 1 - ii;
    const bool act = j <= ii;
    float contrib[AHP];
    float cl = 0.0f;
    const float cj = (j == 0) ? a.lam_pow[ii] : ((j <= ii) ? a.lam_pow[ii - j] * a.one_minus_lam : 0.0f);
    if (act) {
      const float l = -s_rew[t];
      rowl = l + a.gamma * rowl;
      cl = cj * rowl;
    }
#pragma unroll
    for (int ag = 0; ag < AHP; ag += 1) {
      // (1 - gamma) * max_h hs[t][agent], broadcast reads
      if (ag < AH) {
        const int agent = ag / nh;
        float m = s_cost[t * AH + agent * nh];
        for (int h = 1; h < nh; ++h) m = nanmax(m, s_cost[t * AH + agent * nh + h]);
        if (act) row[ag] = nanmax(s_cost[t * AH + ag], a.one_minus_gamma * m + a.gamma * row[ag]);
        contrib[ag] = act ? cj * row[ag] : 0.0f;
      } else {
        contrib[ag] = 0.0f;
      }
    }
    // workgroup reduction of AH + 1 values: DPP sums inside each 16-lane row (VALU only; a 64-lane __shfl_xor reduction
    // is 6 LDS-crossbar round trips per value), then the 16 row sums of the workgroup meet in LDS
#pragma unroll
    for (int c = 0; c < AHP; ++c) contrib[c] = gae_row16_sum(contrib[c]);
    cl = gae_row16_sum(cl);
    float* part = s_part + (ii & 1) * 16 * (AHP + 1);
    if ((lane & 15) == 0) {
      float* dst = part + (tid >> 4) * (AHP + 1);
#pragma unroll
      for (int c = 0; c < AHP; ++c) dst[c] = contrib[c];
      dst[AHP] = cl;
    }
    // row insertion for the next step (utils.py:53-54)
    if (j == ii + 1) {
#pragma unroll
      for (int c = 0; c < AHP; ++c) row[c] = (c < AH) ? s_Vh[t * AH + c] : 0.0f;
      rowl = s_Vl[t];
    }
    __syncthreads();
    if (tid <= AH) {
      const int c = (tid < AH) ? tid : AHP;
      float q = 0.0f;
#pragma unroll
      for (int rw = 0; rw < 16; ++rw) q += part[rw * (AHP + 1) + c];
      if (tid < AH) a.Qh[((size_t)b * T + t) * AH + tid] = q;
      else a.Ql[(size_t)b * T + t] = q;
    }
  }
}

// Column-parallel variant (the default for lambda >= 0.5, T <= 256): the DP's rows never interact except in the final
// lambda-weighted sum, and its columns (n*nh constraint values + the cost value) never interact at all.  So NRG adjacent
// lanes own one (env, column) pair, each keeps RPG consecutive DP rows of that column IN REGISTERS, and a step is RPG x
// (fma, max, select, fma) of straight-line code per lane plus a log2(NRG)-step DPP sum — no LDS, no barrier (the row-parallel
// kernel above spends its time in a 17-value workgroup reduction per step: 1.84 ms per call at B = 4096, T = 128).
//   * weights: c_j(ii) = lambda^(ii-j) (1-lambda)  =  lambda^(ii - j0) * [lambda^-(j-j0) (1-lambda)]  with j0 the lane's first row:
//     the bracket is a per-register constant (lambda^-r, r < RPG <= 32, bounded by 2^31 for lambda >= 0.5), the lane keeps
//     q = sum_r w_r row_r and a running factor f = lambda^(ii - j0) (one multiply per step); row 0's weight is 1 instead of 1-lambda.
//   * inactive rows (j > ii) are kept at exactly 0 by the select, so they add 0 (also with NaN inputs elsewhere).
//   * the step loop is unrolled over one row group (RPG steps), so the row that is inserted after step ii (j = ii + 1) is a
//     compile-time register index; the lane that owns it is chosen by a comparison with the chunk index.
// Same recurrences as compute_dec_ocp_gae (algo/utils.py:11-79); only the summation order of Q differs (fp32, ~1e-7).
template <int NRG> __device__ inline float rg_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
  if (NRG == 8) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, false));   // row_half_mirror
  return v;
}

template <int RPG, int NRG>
__global__ void __launch_bounds__(256) gae_cols_kernel(GaeArgs a) {
  const int T = a.T, AH = a.AH, nh = a.nh, C = AH + 1;
  const int rg = threadIdx.x % NRG;
  const long item = (long)blockIdx.x * (256 / NRG) + threadIdx.x / NRG;
  const bool valid = item < (long)a.B * C;
  const long it = valid ? item : 0;
  const int b = (int)(it / C), col = (int)(it - (long)b * C);
  const bool is_l = (col == AH);
  const int agent = is_l ? 0 : col / nh;
  const float* costs = a.costs + (size_t)b * T * AH + agent * nh;   // + t*AH + h
  const float* vser = is_l ? a.Vl + (size_t)b * (T + 1) : a.Vh + (size_t)b * (T + 1) * AH + col;   // V[t]: stride vstr
  const int vstr = is_l ? 1 : AH;
  const float* rew = a.rewards + (size_t)b * T;
  const float lam = a.lam_pow[1], inv_lam = 1.0f / lam;
  float w[RPG];                                       // lambda^-r (1 - lambda); row 0 of the whole DP: 1
  {
    float p = a.one_minus_lam;
#pragma unroll
    for (int r = 0; r < RPG; ++r) { w[r] = p; p *= inv_lam; }
    if (rg == 0) w[0] = 1.0f;
  }
  float row[RPG];
#pragma unroll
  for (int r = 0; r < RPG; ++r) row[r] = 0.0f;
  const float vT = vser[(size_t)T * vstr];
  if (rg == 0) row[0] = vT;                           // row 0 <- V(x_T)
  float f = 1.0f;                                     // lambda^(ii - rg*RPG) once the group is active
  // inputs of step t = T-1: lo (the cost itself, or -inf for the cost-value column), k ((1-gamma) max_h cost, or -reward), V[t]
  auto fetch = [&](int t, float& lo, float& k, float& vt) {
    vt = vser[(size_t)t * vstr];
    if (is_l) { lo = -INFINITY; k = -rew[t]; }
    else {
      const float* ct = costs + (size_t)t * AH;
      float m = ct[0];
      for (int h = 1; h < nh; ++h) m = nanmax(m, ct[h]);
      lo = ct[col - agent * nh];
      k = a.one_minus_gamma * m;                      // NaN when any component of this agent's cost is
    }
  };
  float lo, k, vt;
  fetch(T - 1, lo, k, vt);
  // NaN inputs: jnp.maximum propagates NaN (utils.py:39-44) and fmaxf does not.  Instead of a NaN-aware max per DP row, the
  // column carries ONE flag: a NaN cost / reward at step t turns every active row of the column NaN in the reference, row 0
  // is always active, hence Q[t'] is NaN for every t' <= t — and likewise from the step after a NaN value is inserted as a
  // row.  (All lanes of a column group see the same inputs, so the flag is uniform over the group.)
  bool poison = vT != vT;
  const int nchunk = (T + RPG - 1) / RPG;
  for (int chunk = 0; chunk < nchunk; ++chunk) {
    const bool g_le = rg <= chunk, g_lt = rg < chunk;
#pragma unroll
    for (int s = 0; s < RPG; ++s) {
      const int ii = chunk * RPG + s;
      if (ii < T) {                                   // uniform
        const int t = T - 1 - ii;
        float lo_n = 0.0f, k_n = 0.0f, vt_n = 0.0f;
        if (t > 0) fetch(t - 1, lo_n, k_n, vt_n);     // the next step's inputs fly under this step's arithmetic
        float q = 0.0f;
#pragma unroll
        for (int r = 0; r < RPG; ++r) {
          const bool act = (r <= s) ? g_le : g_lt;    // row j = rg*RPG + r is active iff j <= ii
          const float nv = fmaxf(lo, fmaf(a.gamma, row[r], k));
          row[r] = act ? nv : row[r];
          q = fmaf(w[r], row[r], q);
        }
        poison = poison || (lo != lo) || (k != k);
        float tot = rg_sum<NRG>(q * f);
        tot = poison ? __builtin_nanf("") : tot;
        if (valid && rg == 0) {
          if (is_l) a.Ql[(size_t)b * T + t] = tot;
          else a.Qh[((size_t)b * T + t) * AH + col] = tot;
        }
        poison = poison || (vt != vt);                // V[t] becomes row ii + 1: part of every sum from the next step on
        if (g_le) f *= lam;
        // row insertion for the next step: j = ii + 1 <- V[t]   (utils.py:53-54)
        const int rn = (s + 1) % RPG;                 // constant after unrolling: the select below touches one register
        const bool mine = rg == chunk + ((s + 1) / RPG);
#pragma unroll
        for (int r = 0; r < RPG; ++r)
          if (r == rn) row[r] = mine ? vt : row[r];
        lo = lo_n; k = k_n; vt = vt_n;
      }
    }
  }
}

template <int RPG, int NRG>
static void launch_gae_cols(const GaeArgs& a, hipStream_t st) {
  const long items = (long)a.B * (a.AH + 1);
  const int per_block = 256 / NRG;
  hipLaunchKernelGGL((gae_cols_kernel<RPG, NRG>), dim3((unsigned)((items + per_block - 1) / per_block)), dim3(256), 0, st, a);
}

extern "C" int32_t dgppo_gae(const float* costs, const float* rewards, const float* Vh, const float* Vl,
                             const float* lam_pow, float gamma, float one_minus_gamma, float one_minus_lam, float* Qh,
                             float* Ql, int32_t B, int32_t T, int32_t n, int32_t nh, void* stream) {
  DGPPO_REQUIRE(B >= 0 && T >= 1 && n >= 1 && nh >= 1, "gae: bad sizes");
  if (B == 0) return 0;
  DGPPO_REQUIRE(costs && rewards && Vh && Vl && lam_pow && Qh && Ql, "gae: NULL operand");
  GaeArgs a{costs, rewards, Vh, Vl, lam_pow, Qh, Ql, B, T, n * nh, n, nh, gamma, one_minus_gamma, one_minus_lam};
  // column-parallel kernel: lambda^-31 must stay in range (lambda = 1 - one_minus_lam >= 0.5) and the rows must fit the lanes
  if (T <= 256 && one_minus_lam <= 0.5f && one_minus_lam >= 0.0f && !getenv("DGPPO_GAE_ROWS")) {
    hipStream_t st = (hipStream_t)stream;
    if (T <= 32) launch_gae_cols<8, 4>(a, st);
    else if (T <= 64) launch_gae_cols<16, 4>(a, st);
    else if (T <= 128) launch_gae_cols<32, 4>(a, st);
    else launch_gae_cols<32, 8>(a, st);
    DGPPO_LAUNCH_CHECK();
    return 0;
  }
  if (T + 1 <= 256 && a.AH <= 32) {
    const int ahp = a.AH <= 8 ? 8 : (a.AH <= 16 ? 16 : 32);
    const size_t fsm = sizeof(float) * ((size_t)T * a.AH + (size_t)(T + 1) * a.AH + (T + 1) + T + 2 * 16 * (ahp + 1));
    if (fsm <= 64 * 1024) {
      hipStream_t st = (hipStream_t)stream;
      if (ahp == 8) hipLaunchKernelGGL(gae_rows_kernel<8>, dim3(B), dim3(256), fsm, st, a);
      else if (ahp == 16) hipLaunchKernelGGL(gae_rows_kernel<16>, dim3(B), dim3(256), fsm, st, a);
      else hipLaunchKernelGGL(gae_rows_kernel<32>, dim3(B), dim3(256), fsm, st, a);
      DGPPO_LAUNCH_CHECK();
      return 0;
    }
  }
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
  const float* Vh = a.Vh ? a.Vh + (size_t)b * (T + 1) * n * nh : nullptr;
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
    for (int h = 0; a.Vh != nullptr && h < nh; ++h) {       // Vh == NULL: plain normalised advantage (InforMARL)
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
  DGPPO_REQUIRE(Ql && Vl && adv && stats, "advantage: NULL operand");   // Vh may be NULL (no CBF terms)
  AdvArgs a{Ql, Vl, Vh, adv, stats, B, T, n, nh, 1.0f / dt, alpha, cbf_eps, cbf_weight};
  hipLaunchKernelGGL(adv_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, a);
  DGPPO_LAUNCH_CHECK();
  return 0;
}

// InforMARL's stage cost (dgppo/algo/informarl.py:329): l = -reward + w * sum_{agents, components} max(cost, 0), returned as
// the equivalent reward r' = reward - w * sum(...) so that dgppo_gae (l = -r') applies unchanged.
__global__ void shaped_reward_kernel(const float* __restrict__ reward, const float* __restrict__ cost, float w,
                                     float* __restrict__ out, long rows, int n, int nh) {
  const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const float* c = cost + r * n * nh;
  float tot = 0.0f;
  for (int i = 0; i < n; ++i) {                    // .sum(axis=-1).sum(axis=-1): components first, then agents
    float si = 0.0f;
    for (int h = 0; h < nh; ++h) si += fmaxf(c[i * nh + h], 0.0f);
    tot += si;
  }
  out[r] = reward[r] - w * tot;
}

extern "C" int32_t dgppo_shaped_reward(const float* reward, const float* cost, float cost_weight, float* out, int64_t rows,
                                       int32_t n, int32_t nh, void* stream) {
  DGPPO_REQUIRE(rows >= 0 && n >= 1 && nh >= 1, "shaped_reward: bad sizes");
  if (rows == 0) return 0;
  DGPPO_REQUIRE(reward && cost && out, "shaped_reward: NULL operand");
  hipLaunchKernelGGL(shaped_reward_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, reward,
                     cost, cost_weight, out, (long)rows, n, nh);
  DGPPO_LAUNCH_CHECK();
  return 0;
}

// =====================================================================================================================
// InforMARL-Lagrangian (dgppo/algo/informarl_lagr.py): advantage with the per-(agent, component) multipliers and the
// multiplier update.
// =====================================================================================================================
struct AdvLagrArgs {
  const float* Ql; const float* Vl; const float* Qh; const float* Vh; const float* lagr;
  float* adv;       // [B,T,n]
  float* Ah;        // [B,T,n,nh] normalised constraint advantage (update_lagr reads it again)
  int B, T, n, nh;
};

// informarl_lagr.py:219-235: Al = (Ql - Vl) standardised per env over T, negated; Ah = (Qh - Vh) standardised per
// (env, agent, component) over T; A = -Al - mean_h(Ah * lagr[a, h]).  One workgroup per env; population std, + 1e-8.
__global__ void __launch_bounds__(256) adv_lagr_kernel(AdvLagrArgs a) {
  __shared__ float red[256];
  __shared__ float s_mean[1 + 64], s_den[1 + 64];
  const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const int T = a.T, n = a.n, nh = a.nh, AH = n * nh;
  const float* Ql = a.Ql + (size_t)b * T;
  const float* Vl = a.Vl + (size_t)b * (T + 1);
  const float* Qh = a.Qh + (size_t)b * T * AH;
  const float* Vh = a.Vh + (size_t)b * (T + 1) * AH;
  // series 0: Ql - Vl ; series 1 + c: Qh - Vh of column c = agent * nh + h
  for (int s0 = 0; s0 < 1 + AH; ++s0) {
    float acc = 0.0f;
    for (int t = tid; t < T; t += nt) acc += (s0 == 0) ? (Ql[t] - Vl[t]) : (Qh[(size_t)t * AH + s0 - 1] - Vh[(size_t)t * AH + s0 - 1]);
    red[tid] = acc;
    __syncthreads();
    for (int o = nt / 2; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    const float mean = red[0] / (float)T;
    __syncthreads();
    acc = 0.0f;
    for (int t = tid; t < T; t += nt) {
      const float d = ((s0 == 0) ? (Ql[t] - Vl[t]) : (Qh[(size_t)t * AH + s0 - 1] - Vh[(size_t)t * AH + s0 - 1])) - mean;
      acc += d * d;
    }
    red[tid] = acc;
    __syncthreads();
    for (int o = nt / 2; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    if (tid == 0) { s_mean[s0] = mean; s_den[s0] = sqrtf(red[0] / (float)T) + 1e-8f; }
    __syncthreads();
  }
  for (int idx = tid; idx < T * n; idx += nt) {
    const int t = idx / n, ag = idx - t * n;
    const float al = -(((Ql[t] - Vl[t]) - s_mean[0]) / s_den[0]);
    float acc = 0.0f;
    for (int h = 0; h < nh; ++h) {
      const int c = ag * nh + h;
      const float ah = ((Qh[(size_t)t * AH + c] - Vh[(size_t)t * AH + c]) - s_mean[1 + c]) / s_den[1 + c];
      a.Ah[((size_t)b * T + t) * AH + c] = ah;
      acc += ah * a.lagr[c];
    }
    a.adv[(size_t)b * T * n + idx] = al - acc / (float)nh;
  }
}

extern "C" int32_t dgppo_advantage_lagr(const float* Ql, const float* Vl, const float* Qh, const float* Vh,
                                        const float* lagr, float* adv, float* Ah, int32_t B, int32_t T, int32_t n,
                                        int32_t nh, void* stream) {
  DGPPO_REQUIRE(B >= 0 && T >= 1 && n >= 1 && nh >= 1 && n * nh <= 64, "advantage_lagr: bad sizes (n * nh <= 64)");
  if (B == 0) return 0;
  DGPPO_REQUIRE(Ql && Vl && Qh && Vh && lagr && adv && Ah, "advantage_lagr: NULL operand");
  AdvLagrArgs a{Ql, Vl, Qh, Vh, lagr, adv, Ah, B, T, n, nh};
  hipLaunchKernelGGL(adv_lagr_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, a);
  DGPPO_LAUNCH_CHECK();
  return 0;
}

// update_lagr (informarl_lagr.py:286-309): delta[a,h] = -mean_{b,t}( Vh (1 - gamma) + exp(log_pi_new - log_pi_old) Ah ),
// lagr = relu(lagr - delta * lr).  rows = b*T*n entries, agent = row % n; Vh is addressed with its own env stride
// (it is the [B, T+1, n, nh] value array, of which the first T steps of each env are used).
__global__ void __launch_bounds__(256) lagr_sum_kernel(const float* __restrict__ lp_new, const float* __restrict__ lp_old,
                                                       const float* __restrict__ Vh, const float* __restrict__ Ah,
                                                       float* __restrict__ sums, long rows, int n, int nh, int T,
                                                       long vh_env_stride, float one_minus_gamma) {
  extern __shared__ float s_acc[];   // [n * nh]
  for (int c = threadIdx.x; c < n * nh; c += blockDim.x) s_acc[c] = 0.0f;
  __syncthreads();
  for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (long)gridDim.x * blockDim.x) {
    const int ag = (int)(r % n);
    const long bt = r / n, e = bt / T, t = bt - e * T;
    const float ratio = expf(lp_new[r] - lp_old[r]);
    for (int h = 0; h < nh; ++h) {
      const float vh = Vh[e * vh_env_stride + (t * n + ag) * nh + h];
      atomicAdd(&s_acc[ag * nh + h], vh * one_minus_gamma + ratio * Ah[r * nh + h]);
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < n * nh; c += blockDim.x) atomicAdd(&sums[c], s_acc[c]);
}

__global__ void lagr_apply_kernel(float* __restrict__ lagr, float* __restrict__ sums, int count, float inv_bt, float lr) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= count) return;
  const float delta = -(sums[c] * inv_bt);
  lagr[c] = fmaxf(lagr[c] - delta * lr, 0.0f);
  sums[c] = 0.0f;                        // ready for the next minibatch
}

extern "C" int32_t dgppo_lagr_update(const float* lp_new, const float* lp_old, const float* Vh, int64_t vh_env_stride,
                                     const float* Ah, float* lagr, float* sums, int32_t n_env, int32_t T, int32_t n,
                                     int32_t nh, float one_minus_gamma, float lr, void* stream) {
  DGPPO_REQUIRE(n_env >= 1 && T >= 1 && n >= 1 && nh >= 1 && n * nh <= 1024, "lagr_update: bad sizes");
  DGPPO_REQUIRE(lp_new && lp_old && Vh && Ah && lagr && sums, "lagr_update: NULL operand");
  const long rows = (long)n_env * T * n;
  const int grid = (int)((rows + 255) / 256 < 512 ? (rows + 255) / 256 : 512);
  hipLaunchKernelGGL(lagr_sum_kernel, dim3(grid), dim3(256), sizeof(float) * n * nh, (hipStream_t)stream, lp_new, lp_old, Vh,
                     Ah, sums, rows, n, nh, T, (long)vh_env_stride, one_minus_gamma);
  hipLaunchKernelGGL(lagr_apply_kernel, dim3((n * nh + 63) / 64), dim3(64), 0, (hipStream_t)stream, lagr, sums, n * nh,
                     1.0f / (float)((long)n_env * T), lr);
  DGPPO_LAUNCH_CHECK();
  return 0;
}

// the two halves of dgppo_lagr_update for the data-parallel update: every rank accumulates the sums of ITS share of the
// minibatch, the caller all-reduces sums[n*nh], then every rank applies the identical step with the GLOBAL row count
extern "C" int32_t dgppo_lagr_sums(const float* lp_new, const float* lp_old, const float* Vh, int64_t vh_env_stride,
                                   const float* Ah, float* sums, int32_t n_env, int32_t T, int32_t n, int32_t nh,
                                   float one_minus_gamma, void* stream) {
  DGPPO_REQUIRE(n_env >= 1 && T >= 1 && n >= 1 && nh >= 1 && n * nh <= 1024, "lagr_sums: bad sizes");
  DGPPO_REQUIRE(lp_new && lp_old && Vh && Ah && sums, "lagr_sums: NULL operand");
  const long rows = (long)n_env * T * n;
  const int grid = (int)((rows + 255) / 256 < 512 ? (rows + 255) / 256 : 512);
  hipLaunchKernelGGL(lagr_sum_kernel, dim3(grid), dim3(256), sizeof(float) * n * nh, (hipStream_t)stream, lp_new, lp_old, Vh,
                     Ah, sums, rows, n, nh, T, (long)vh_env_stride, one_minus_gamma);
  DGPPO_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t dgppo_lagr_apply(float* lagr, float* sums, int32_t count, int64_t rows_total, float lr, void* stream) {
  DGPPO_REQUIRE(count >= 1 && count <= 1024 && rows_total >= 1, "lagr_apply: bad sizes");
  DGPPO_REQUIRE(lagr && sums, "lagr_apply: NULL operand");
  hipLaunchKernelGGL(lagr_apply_kernel, dim3((count + 63) / 64), dim3(64), 0, (hipStream_t)stream, lagr, sums, count,
                     1.0f / (float)rows_total, lr);
  DGPPO_LAUNCH_CHECK();
  return 0;
}

// out = max(x, 0) elementwise: the clipped costs fed to the Dec-OCP GAE by the Lagrangian baseline
// (jnp.clip(rollout.costs, a_min=0), informarl_lagr.py:213)
__global__ void relu_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, long count) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) out[i] = fmaxf(x[i], 0.0f);
}

extern "C" int32_t dgppo_relu_fwd(const float* x, float* out, int64_t count, void* stream) {
  DGPPO_REQUIRE(count >= 0, "relu_fwd: count < 0");
  if (count == 0) return 0;
  DGPPO_REQUIRE(x && out, "relu_fwd: NULL operand");
  hipLaunchKernelGGL(relu_fwd_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, out,
                     (long)count);
  DGPPO_LAUNCH_CHECK();
  return 0;
}
