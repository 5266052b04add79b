// Row-wise network pieces (fp32): LayerNorm+ReLU, GRU scan (MFMA recurrent product), tanh-Normal head, agent mean-pool,
// PPO / value losses — forward and hand-written backward.
//
// Reference arithmetic replaced (file:line relative to /root/reference):
//   MLP block Dense->LayerNorm->ReLU     dgppo/nn/mlp.py:17-29 (flax nn.LayerNorm, eps 1e-6, fast variance)
//   RNN / flax GRUCell                   dgppo/nn/rnn.py:14-30
//   TanhNormal head + distribution       dgppo/algo/module/policy.py:62-74,191-212 ; distribution.py:10-46
//   RStateFn mean pool                   dgppo/algo/module/value.py:31-33
//   losses                               dgppo/algo/informarl.py:374,428-438 ; dgppo/algo/dgppo.py:310
//   backward = jax.grad of the above     dgppo/algo/informarl.py:377,440 ; dgppo/algo/dgppo.py:316
#include "common.h"
#include <stdlib.h>

using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// Sum of NV per-thread values over a 256-thread workgroup, then ONE atomicAdd per value.  Scalar statistics used to be
// added with one atomic per WAVE: thousands of float atomics on the same address serialise in L2 (~7 ns each; 30 us for
// value_loss at 262 144 elements); with a grid-stride kernel of <= 256 workgroups it is 256 atomics per value.
template <int NV>
__device__ inline void block_atomic_sums(const float (&v)[NV], float* dst) {
  __shared__ float s_red[4][NV];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const float w = wave_sum(v[k]);
    if (lane == 0) s_red[wave][k] = w;
  }
  __syncthreads();
  if (threadIdx.x < NV) {
    float t = 0.0f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += s_red[w][threadIdx.x];
    atomicAdd(dst + threadIdx.x, t);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// LayerNorm(64) + ReLU: four rows per wave — the 16 lanes of a DPP row own one matrix row, 4 consecutive columns each
// (one float4).  Row sums = 3 in-lane adds + 4 DPP steps that never leave the VALU (a 64-lane __shfl_xor reduction costs
// 6 LDS-crossbar round trips per sum).
// ---------------------------------------------------------------------------------------------------------------------
__device__ inline float row16_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, false));   // row_half_mirror
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, false));   // row_mirror
  return v;
}

__global__ void __launch_bounds__(256) ln_relu_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ y,
                                                          float* __restrict__ stats, int M) {
  const int lane = threadIdx.x & 63, li = lane & 15, lq = lane >> 4;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nw = (gridDim.x * blockDim.x) >> 6;
  const float4 g = reinterpret_cast<const float4*>(gamma)[li], b = reinterpret_cast<const float4*>(beta)[li];
  for (int row0 = wave * 4; row0 < M; row0 += nw * 4) {
    const int row = row0 + lq;
    const bool ok = row < M;
    const float4 v = reinterpret_cast<const float4*>(x)[(size_t)(ok ? row : M - 1) * 16 + li];
    const float mean = row16_sum((v.x + v.y) + (v.z + v.w)) * (1.0f / 64.0f);
    const float mean2 = row16_sum((v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w)) * (1.0f / 64.0f);
    const float var = fmaxf(mean2 - mean * mean, 0.0f);
    const float rstd = rsqrtf(var + 1e-6f);
    float4 o;
    o.x = fmaxf((v.x - mean) * rstd * g.x + b.x, 0.0f);
    o.y = fmaxf((v.y - mean) * rstd * g.y + b.y, 0.0f);
    o.z = fmaxf((v.z - mean) * rstd * g.z + b.z, 0.0f);
    o.w = fmaxf((v.w - mean) * rstd * g.w + b.w, 0.0f);
    if (ok) {
      reinterpret_cast<float4*>(y)[(size_t)row * 16 + li] = o;
      if (stats != nullptr && li == 0) reinterpret_cast<float2*>(stats)[row] = make_float2(mean, rstd);
    }
  }
}

// dx = rstd * (dxhat - mean(dxhat) - xhat * mean(dxhat * xhat)),  dxhat = dy * (y > 0) * gamma
__global__ void __launch_bounds__(256) ln_relu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                          const float* __restrict__ stats, const float* __restrict__ gamma,
                                                          const float* __restrict__ dy, float* __restrict__ dx,
                                                          float* __restrict__ dgamma, float* __restrict__ dbeta, int M) {
  const int lane = threadIdx.x & 63, li = lane & 15, lq = lane >> 4;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nw = (gridDim.x * blockDim.x) >> 6;
  const float4 g = reinterpret_cast<const float4*>(gamma)[li];
  float dg[4] = {0.f, 0.f, 0.f, 0.f}, db[4] = {0.f, 0.f, 0.f, 0.f};
  for (int row0 = wave * 4; row0 < M; row0 += nw * 4) {
    const int row = row0 + lq;
    const bool ok = row < M;
    const size_t o = (size_t)(ok ? row : M - 1) * 16 + li;
    const float4 xv = reinterpret_cast<const float4*>(x)[o], yv = reinterpret_cast<const float4*>(y)[o];
    const float4 dv = reinterpret_cast<const float4*>(dy)[o];
    const float2 st = reinterpret_cast<const float2*>(stats)[ok ? row : M - 1];
    const float mean = st.x, rstd = st.y;
    const float xh[4] = {(xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd};
    const float dl[4] = {(ok && yv.x > 0.0f) ? dv.x : 0.0f, (ok && yv.y > 0.0f) ? dv.y : 0.0f,
                         (ok && yv.z > 0.0f) ? dv.z : 0.0f, (ok && yv.w > 0.0f) ? dv.w : 0.0f};
    const float gg[4] = {g.x, g.y, g.z, g.w};
    float dxh[4], s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      dg[q] += dl[q] * xh[q];
      db[q] += dl[q];
      dxh[q] = dl[q] * gg[q];
      s1 += dxh[q];
      s2 += dxh[q] * xh[q];
    }
    const float m1 = row16_sum(s1) * (1.0f / 64.0f), m2 = row16_sum(s2) * (1.0f / 64.0f);
    if (ok)
      reinterpret_cast<float4*>(dx)[o] = make_float4(rstd * (dxh[0] - m1 - xh[0] * m2), rstd * (dxh[1] - m1 - xh[1] * m2),
                                                     rstd * (dxh[2] - m1 - xh[2] * m2), rstd * (dxh[3] - m1 - xh[3] * m2));
  }
  // per-workgroup reduction of the 16 (wave, row group) partial sums, then one atomic per column
  __shared__ float red[2][16][64];
  const int slot = (threadIdx.x >> 6) * 4 + lq;
#pragma unroll
  for (int q = 0; q < 4; ++q) { red[0][slot][li * 4 + q] = dg[q]; red[1][slot][li * 4 + q] = db[q]; }
  __syncthreads();
  if (threadIdx.x < 128) {
    const int which = threadIdx.x >> 6;
    float v = 0.0f;
#pragma unroll
    for (int sidx = 0; sidx < 16; ++sidx) v += red[which][sidx][lane];
    atomicAdd((which ? dbeta : dgamma) + lane, v);
  }
}

extern "C" int32_t dgppo_ln_relu_fwd(const float* x, const float* gamma, const float* beta, float* y, float* stats,
                                     int32_t M, void* stream) {
  DGPPO_REQUIRE(M >= 0, "ln_relu_fwd: M < 0");
  if (M == 0) return 0;
  DGPPO_REQUIRE(x && gamma && beta && y, "ln_relu_fwd: NULL operand");
  const int grid = min(cdiv(M, 16), 2048);
  hipLaunchKernelGGL(ln_relu_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y, stats, M);
  DGPPO_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t dgppo_ln_relu_bwd(const float* x, const float* y, const float* stats, const float* gamma,
                                     const float* dy, float* dx, float* dgamma, float* dbeta, int32_t M, void* stream) {
  DGPPO_REQUIRE(M >= 0, "ln_relu_bwd: M < 0");
  if (M == 0) return 0;
  DGPPO_REQUIRE(x && y && stats && gamma && dy && dx && dgamma && dbeta, "ln_relu_bwd: NULL operand");
  // every workgroup ends with 128 float atomics on the same two cache lines (dgamma / dbeta): those serialise in L2, so the
  // grid is kept at what the streaming part needs (DGPPO_LN_BWD_GRID: tuning override)
  static const int cap = getenv("DGPPO_LN_BWD_GRID") ? atoi(getenv("DGPPO_LN_BWD_GRID")) : 256;
  const int grid = min(cdiv(M, 16), cap);
  hipLaunchKernelGGL(ln_relu_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, y, stats, gamma, dy, dx, dgamma,
                     dbeta, M);
  DGPPO_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// GRU scan.  Sequence s (0 <= s < n_seq) at step tau lives in row  ((s / n_inner) * T + tau) * n_inner + (s % n_inner)
// of every [rows, *] activation matrix (n_inner = agents per graph, so graphs are ordered (env, time) and a chunk of T
// consecutive graphs of one env forms the time axis).  gi = x W_i + b_i is precomputed for all rows by the dense kernel.
//   r = sig(gi_r + h Whr)  z = sig(gi_z + h Whz)  n = tanh(gi_n + r * (h Whn + bhn))  h' = (1 - z) n + z h
// ---------------------------------------------------------------------------------------------------------------------
#define GRU_H 64
#define GRU_WL 194   // padded LDS row of the gate gradients: stride = 2 (mod 32) as well
#define GRU_HL 66    // padded LDS row of h: stride = 2 (mod 32), conflict-free A-fragment reads (see nn_dense.hip)

struct GruArgs {
  const float* gi;      // [rows, 192]
  const float* Wh;      // [64, 192]  (cols r | z | n)
  const float* bhn;     // [64]
  const float* h0;      // [n_seq, 64] or NULL (zeros)
  float* hs;            // [rows, 64]  h after each step
  float* hprev;         // [rows, 64]  h before each step (saved for backward) or NULL
  float* gates;         // [rows, 256] r | z | n | hn_lin (saved for backward) or NULL
  // backward
  const float* dhs;     // [rows, 64] gradient w.r.t. every step's output
  float* dgi;           // [rows, 192]
  float* dgh;           // [rows, 192]  (da_r | da_z | d hn_lin) -> dWh = hprev^T dgh, dbhn = colsum(dgh[:,128:])
  int n_seq, T, n_inner;
};

__device__ inline size_t gru_row(const GruArgs& a, int s, int tau) {
  const int grp = s / a.n_inner, i = s - grp * a.n_inner;
  return ((size_t)grp * a.T + tau) * a.n_inner + i;
}
__device__ inline float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
// GRU gate non-linearities on the hardware transcendental units: exp through v_exp_f32 (scaled by log2 e), the quotient
// through v_rcp_f32; absolute error ~1e-7 on outputs in [-1, 1], far inside the 1e-5 parity tolerance, at ~1/4 of the
// VALU instructions of expf / tanhf / IEEE division (the gate math was ~70 instructions per element).
__device__ inline float gate_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ inline float gate_tanh(float x) { return fmaf(2.0f, __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * x)), -1.0f); }


#define GRU_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// Persistent workgroups walk tiles of RB = 16*RTW sequences.  Wave w owns hidden columns 16w..16w+15 of all three gates
// (column tiles w, w+4, w+8 of h Wh), and its Wh fragments (16 k-steps x 3 tiles = 48 registers) stay resident for the
// whole launch.  The h tile is double-buffered in LDS: one LDS-only barrier per step.  The gate inputs gi of a step are
// requested before its MFMAs and consumed after them; the next tile's h0 rows are requested at the start of a tile.
template <int RTW>
__global__ void __launch_bounds__(256) gru_fwd_kernel(GruArgs a) {
  extern __shared__ float sm[];
  constexpr int RB = 16 * RTW;
  constexpr int HSLOT = RB * 16 / 256 > 0 ? RB * 16 / 256 : 1;   // float4 slots of an h tile per lane (RB*16 / 256)
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 15, lq = lane >> 4;
  const int c = w * 16 + li;                                     // hidden column of this lane
  float breg[16][3];
#pragma unroll
  for (int kk = 0; kk < 16; ++kk)
#pragma unroll
    for (int t = 0; t < 3; ++t) breg[kk][t] = a.Wh[(size_t)(kk * 4 + lq) * 192 + t * 64 + c];
  const float bn = a.bhn[c];
  const int n_tiles = (a.n_seq + RB - 1) / RB;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float hpf[HSLOT][4];                   // (plain floats: a float4 array captured by the lambdas ends up in scratch)
  auto fetch_h0 = [&](int tile) {
#pragma unroll
    for (int u = 0; u < HSLOT; ++u) {
      int idx = u * 256 + tid;
      idx = idx < RB * 16 ? idx : RB * 16 - 1;
      const int r = idx >> 4, q = idx & 15;
      int s = tile * RB + r;
      s = s < a.n_seq ? s : a.n_seq - 1;
      float4 v = z4;
      if (a.h0 != nullptr) v = reinterpret_cast<const float4*>(a.h0 + (size_t)s * GRU_H)[q];
      hpf[u][0] = v.x; hpf[u][1] = v.y; hpf[u][2] = v.z; hpf[u][3] = v.w;
    }
  };
  auto commit_h0 = [&](float* dst) {
#pragma unroll
    for (int u = 0; u < HSLOT; ++u) {
      const int idx = u * 256 + tid, r = idx >> 4, q = idx & 15;
      if (idx < RB * 16) { float* d = dst + r * GRU_HL + 4 * q; d[0] = hpf[u][0]; d[1] = hpf[u][1]; d[2] = hpf[u][2]; d[3] = hpf[u][3]; }
    }
  };
  int tile = blockIdx.x;
  if (tile < n_tiles) { fetch_h0(tile); commit_h0(sm); }
  __syncthreads();
  int cur = 0;
  for (; tile < n_tiles; tile += gridDim.x) {
    const int s0 = tile * RB;
    const int nxt = tile + gridDim.x;
    if (nxt < n_tiles) fetch_h0(nxt);
    // rows of this lane's output elements at tau = 0 (row(tau) = row0 + tau * n_inner), clamped for loads
    int row0[RTW][4];
    bool ok[RTW][4];
#pragma unroll
    for (int rt = 0; rt < RTW; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int s = s0 + rt * 16 + lq * 4 + r;
        ok[rt][r] = s < a.n_seq;
        s = ok[rt][r] ? s : a.n_seq - 1;
        // row(s, tau = 0); with T == 1 (or one sequence per group) rows and sequences coincide: no integer division
        int rw = s;
        if (a.T != 1 && a.n_inner != 1) { const int grp = s / a.n_inner; rw = grp * a.T * a.n_inner + (s - grp * a.n_inner); }
        else if (a.T != 1) rw = s * a.T;
        row0[rt][r] = rw;
      }
    // the input-gate rows of step tau + 1 are requested while step tau computes (one step of latency hiding: the product of
    // a step is ~1.5 k cycles, a global load under load more)
    float gn[RTW][4][3];
    auto fetch_gi = [&](int tau) {
#pragma unroll
      for (int rt = 0; rt < RTW; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float* gp = a.gi + (size_t)(row0[rt][r] + tau * a.n_inner) * 192 + c;
          gn[rt][r][0] = gp[0]; gn[rt][r][1] = gp[64]; gn[rt][r][2] = gp[128];
        }
    };
    fetch_gi(0);
    for (int tau = 0; tau < a.T; ++tau) {
      const float* hs_cur = sm + cur * (RB * GRU_HL);
      float* hs_nxt = sm + (cur ^ 1) * (RB * GRU_HL);
      float g[RTW][4][3];
#pragma unroll
      for (int rt = 0; rt < RTW; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { g[rt][r][0] = gn[rt][r][0]; g[rt][r][1] = gn[rt][r][1]; g[rt][r][2] = gn[rt][r][2]; }
      if (tau + 1 < a.T) fetch_gi(tau + 1);
      __builtin_amdgcn_sched_barrier(0);
      float areg[RTW][16];
#pragma unroll
      for (int rt = 0; rt < RTW; ++rt)
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) areg[rt][kk] = hs_cur[(rt * 16 + li) * GRU_HL + kk * 4 + lq];
      __builtin_amdgcn_sched_barrier(0);
      f32x4 acc[RTW][3];
#pragma unroll
      for (int rt = 0; rt < RTW; ++rt)
#pragma unroll
        for (int t = 0; t < 3; ++t) acc[rt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 16; ++kk)
#pragma unroll
        for (int rt = 0; rt < RTW; ++rt)
#pragma unroll
          for (int t = 0; t < 3; ++t) acc[rt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[rt][kk], breg[kk][t], acc[rt][t], 0, 0, 0);
#pragma unroll
      for (int rt = 0; rt < RTW; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int rl = rt * 16 + lq * 4 + r;
          const float hp = hs_cur[rl * GRU_HL + c];
          const float rg = gate_sigmoid(g[rt][r][0] + acc[rt][0][r]);
          const float zg = gate_sigmoid(g[rt][r][1] + acc[rt][1][r]);
          const float hn = acc[rt][2][r] + bn;
          const float ng = gate_tanh(g[rt][r][2] + rg * hn);
          const float hnew = (1.0f - zg) * ng + zg * hp;
          hs_nxt[rl * GRU_HL + c] = hnew;
          if (ok[rt][r]) {
            const size_t row = (size_t)(row0[rt][r] + tau * a.n_inner);
            a.hs[row * GRU_H + c] = hnew;
            if (a.hprev != nullptr) a.hprev[row * GRU_H + c] = hp;
            if (a.gates != nullptr) {
              float* gs = a.gates + row * 256;
              gs[c] = rg; gs[64 + c] = zg; gs[128 + c] = ng; gs[192 + c] = hn;
            }
          }
        }
      GRU_LDS_BARRIER();
      cur ^= 1;
    }
    if (nxt < n_tiles) {
      commit_h0(sm + cur * (RB * GRU_HL));
      GRU_LDS_BARRIER();
    }
  }
}

// BPTT: walks tau = T-1 .. 0.  Each lane owns the same (sequence row, hidden column) elements as in the forward (the
// MFMA output layout), so the carried dh lives in registers: dh_prev = dh * z + (dgh Wh^T), the product running on the
// matrix cores with the Wh^T fragments (48 k-steps x 1 column tile) resident in registers.  dgh tiles are
// double-buffered in LDS: one LDS-only barrier per step.  The saved gates / hprev / upstream gradient of step tau-1 are
// requested while step tau's product runs.
template <int RTW>
__global__ void __launch_bounds__(256) gru_bwd_kernel(GruArgs a) {
  extern __shared__ float sm[];
  constexpr int RB = 16 * RTW;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 15, lq = lane >> 4;
  const int c = w * 16 + li;
  float breg[48];                       // B[k][j = c] = Wh[c][k]
#pragma unroll
  for (int kk = 0; kk < 48; ++kk) breg[kk] = a.Wh[(size_t)c * 192 + kk * 4 + lq];
  const int n_tiles = (a.n_seq + RB - 1) / RB;
  int cur = 0;
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int s0 = tile * RB;
    int row0[RTW][4];
    bool ok[RTW][4];
#pragma unroll
    for (int rt = 0; rt < RTW; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int s = s0 + rt * 16 + lq * 4 + r;
        ok[rt][r] = s < a.n_seq;
        s = ok[rt][r] ? s : a.n_seq - 1;
        // row(s, tau = 0); with T == 1 (or one sequence per group) rows and sequences coincide: no integer division
        int rw = s;
        if (a.T != 1 && a.n_inner != 1) { const int grp = s / a.n_inner; rw = grp * a.T * a.n_inner + (s - grp * a.n_inner); }
        else if (a.T != 1) rw = s * a.T;
        row0[rt][r] = rw;
      }
    float rg[RTW][4], zg[RTW][4], ng[RTW][4], hn[RTW][4], hp[RTW][4], du[RTW][4], dh[RTW][4];
    auto fetch = [&](int tau) {
#pragma unroll
      for (int rt = 0; rt < RTW; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const size_t row = (size_t)(row0[rt][r] + tau * a.n_inner);
          const float* gs = a.gates + row * 256 + c;
          rg[rt][r] = gs[0]; zg[rt][r] = gs[64]; ng[rt][r] = gs[128]; hn[rt][r] = gs[192];
          hp[rt][r] = a.hprev[row * GRU_H + c];
          du[rt][r] = a.dhs[row * GRU_H + c];
        }
    };
#pragma unroll
    for (int rt = 0; rt < RTW; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) dh[rt][r] = 0.0f;
    fetch(a.T - 1);
    for (int tau = a.T - 1; tau >= 0; --tau) {
      float* s_g = sm + cur * (RB * GRU_WL);
      // phase A: gate gradients of this lane's elements
#pragma unroll
      for (int rt = 0; rt < RTW; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int rl = rt * 16 + lq * 4 + r;
          const float d = ok[rt][r] ? dh[rt][r] + du[rt][r] : 0.0f;
          const float dz = d * (hp[rt][r] - ng[rt][r]);
          const float dn = d * (1.0f - zg[rt][r]);
          const float dan = dn * (1.0f - ng[rt][r] * ng[rt][r]);
          const float dhn = dan * rg[rt][r];
          const float dar = dan * hn[rt][r] * rg[rt][r] * (1.0f - rg[rt][r]);
          const float daz = dz * zg[rt][r] * (1.0f - zg[rt][r]);
          dh[rt][r] = d * zg[rt][r];
          s_g[rl * GRU_WL + c] = dar;
          s_g[rl * GRU_WL + 64 + c] = daz;
          s_g[rl * GRU_WL + 128 + c] = dhn;
          if (ok[rt][r]) {
            const size_t row = (size_t)(row0[rt][r] + tau * a.n_inner);
            float* o = a.dgi + row * 192 + c;
            o[0] = dar; o[64] = daz; o[128] = dan;
            float* p = a.dgh + row * 192 + c;
            p[0] = dar; p[64] = daz; p[128] = dhn;
          }
        }
      GRU_LDS_BARRIER();
      if (tau > 0) fetch(tau - 1);          // in flight during the product below
      __builtin_amdgcn_sched_barrier(0);
      // phase B: dh_prev[row, c] += sum_k dgh[row, k] * Wh[c, k]
      f32x4 acc[RTW];
#pragma unroll
      for (int rt = 0; rt < RTW; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k0 = 0; k0 < 48; k0 += 16) {
        float areg[RTW][16];
#pragma unroll
        for (int rt = 0; rt < RTW; ++rt)
#pragma unroll
          for (int u = 0; u < 16; ++u) areg[rt][u] = s_g[(rt * 16 + li) * GRU_WL + (k0 + u) * 4 + lq];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
          for (int rt = 0; rt < RTW; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[rt][u], breg[k0 + u], acc[rt], 0, 0, 0);
      }
#pragma unroll
      for (int rt = 0; rt < RTW; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) dh[rt][r] += acc[rt][r];
      cur ^= 1;                             // the next step writes the other dgh buffer: no second barrier needed
    }
    GRU_LDS_BARRIER();                      // (the tile after this one may start on the buffer another wave still reads)
  }
}

// resident workgroups of a kernel on the current device (occupancy x CUs)
static int gru_resident(const void* fn, size_t smem) {
  int per_cu = 0, dev = 0, cus = 256;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, smem) != hipSuccess || per_cu < 1) per_cu = 1;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
  return per_cu * cus;
}

static int32_t gru_check(const GruArgs& a) {
  DGPPO_REQUIRE(a.n_seq >= 0 && a.T >= 1 && a.n_inner >= 1, "gru: bad sizes n_seq=%d T=%d n_inner=%d", a.n_seq, a.T, a.n_inner);
  DGPPO_REQUIRE(a.n_seq % a.n_inner == 0, "gru: n_seq must be a multiple of n_inner");
  return 0;
}

extern "C" int32_t dgppo_gru_fwd(const float* gi, const float* Wh, const float* bhn, const float* h0, float* hs,
                                 float* hprev, float* gates, int32_t n_seq, int32_t T, int32_t n_inner, void* stream) {
  GruArgs a{};
  a.gi = gi; a.Wh = Wh; a.bhn = bhn; a.h0 = h0; a.hs = hs; a.hprev = hprev; a.gates = gates;
  a.n_seq = n_seq; a.T = T; a.n_inner = n_inner;
  int32_t rc = gru_check(a);
  if (rc) return rc;
  if (n_seq == 0) return 0;
  DGPPO_REQUIRE(gi && Wh && bhn && hs, "gru_fwd: NULL operand");
  // 32-sequence tiles when there are enough of them to fill the device, otherwise 16 (half the MFMA chain per step)
  if (cdiv(n_seq, 32) >= 512) {
    const size_t smem = sizeof(float) * 2 * 32 * GRU_HL;
    static thread_local int cap = 0;
    if (cap == 0) cap = gru_resident(reinterpret_cast<const void*>(&gru_fwd_kernel<2>), smem);
    const int tiles = cdiv(n_seq, 32);
    hipLaunchKernelGGL(gru_fwd_kernel<2>, dim3(tiles < cap ? tiles : cap), dim3(256), smem, (hipStream_t)stream, a);
  } else {
    const size_t smem = sizeof(float) * 2 * 16 * GRU_HL;
    static thread_local int cap = 0;
    if (cap == 0) cap = gru_resident(reinterpret_cast<const void*>(&gru_fwd_kernel<1>), smem);
    const int tiles = cdiv(n_seq, 16);
    hipLaunchKernelGGL(gru_fwd_kernel<1>, dim3(tiles < cap ? tiles : cap), dim3(256), smem, (hipStream_t)stream, a);
  }
  DGPPO_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t dgppo_gru_bwd(const float* dhs, const float* Wh, const float* hprev, const float* gates, float* dgi,
                                 float* dgh, int32_t n_seq, int32_t T, int32_t n_inner, void* stream) {
  GruArgs a{};
  a.dhs = dhs; a.Wh = Wh; a.hprev = (float*)hprev; a.gates = (float*)gates; a.dgi = dgi; a.dgh = dgh;
  a.n_seq = n_seq; a.T = T; a.n_inner = n_inner;
  int32_t rc = gru_check(a);
  if (rc) return rc;
  if (n_seq == 0) return 0;
  DGPPO_REQUIRE(dhs && Wh && hprev && gates && dgi && dgh, "gru_bwd: NULL operand");
  if (cdiv(n_seq, 32) >= 512) {
    const size_t smem = sizeof(float) * 2 * 32 * GRU_WL;
    static thread_local int cap = 0;
    if (cap == 0) cap = gru_resident(reinterpret_cast<const void*>(&gru_bwd_kernel<2>), smem);
    const int tiles = cdiv(n_seq, 32);
    hipLaunchKernelGGL(gru_bwd_kernel<2>, dim3(tiles < cap ? tiles : cap), dim3(256), smem, (hipStream_t)stream, a);
  } else {
    const size_t smem = sizeof(float) * 2 * 16 * GRU_WL;
    static thread_local int cap = 0;
    if (cap == 0) cap = gru_resident(reinterpret_cast<const void*>(&gru_bwd_kernel<1>), smem);
    const int tiles = cdiv(n_seq, 16);
    hipLaunchKernelGGL(gru_bwd_kernel<1>, dim3(tiles < cap ? tiles : cap), dim3(256), smem, (hipStream_t)stream, a);
  }
  DGPPO_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// tanh-Normal head (policy.py:62-74, distribution.py:10-46).  ms[row] = [mean0, mean1, std_trans0, std_trans1]
// ---------------------------------------------------------------------------------------------------------------------
#define STD_INIT_INV (-0.43275212956718856f)  // log(exp(0.5) - 1)
#define STD_MIN 1e-5f
#define THRESH 0.999f
#define INV_THRESH 3.8002011672502f           // atanh(0.999)
#define LOG_EPS (-6.907755278982137f)         // log(1 - 0.999)
#define HALF_LOG_2PI 0.9189385332046727f
#define LOG2_F 0.6931471805599453f

__device__ inline float softplusf_(float x) { return (x > 20.0f) ? x : log1pf(expf(x)); }
// log Phi(z): erfc for the bulk, asymptotic series for the far left tail (tfp special_math.log_ndtr)
__device__ inline float log_ndtrf_(float z) {
  if (z > -5.0f) return logf(0.5f * erfcf(-z * 0.70710678118654752f));
  const float z2 = z * z;
  const float series = 1.0f - 1.0f / z2 + 3.0f / (z2 * z2) - 15.0f / (z2 * z2 * z2);
  return -0.5f * z2 - logf(-z) - HALF_LOG_2PI + logf(series);
}
// d/dz log Phi(z) = phi(z) / Phi(z)
__device__ inline float dlog_ndtrf_(float z) {
  if (z > -5.0f) return 0.3989422804014327f * expf(-0.5f * z * z) / (0.5f * erfcf(-z * 0.70710678118654752f));
  const float z2 = z * z;
  return -z / (1.0f - 1.0f / z2 + 3.0f / (z2 * z2));
}
__device__ inline float tanh_fldj(float x) { return 2.0f * (LOG2_F - x - softplusf_(-2.0f * x)); }

struct HeadArgs {
  const float* ms;        // [rows, 4]
  const float* eps;       // sample: [rows, 2] noise; eval: [n_agents, 2] the constant entropy noise eps_hat
  const float* action_in; // eval: [rows, 2]
  float* action;          // sample / mode: [rows, 2]
  float* log_pi;          // [rows]
  float* entropy;         // eval: [rows]
  int rows, n_agents, mode;  // 0 sample, 1 mode, 2 eval
  // PPO loss + backward (eval only)
  const float* log_pi_old;   // [rows]
  const float* adv;          // [rows]
  float* dms;                // [rows, 4]
  float* stats;              // [8]: sum loss_policy, sum entropy, sum(l2 > l1), sum |ratio - 1|, ...
  float clip_eps, coef_ent, inv_count;
};

__device__ inline float tn_log_prob_dim(float act, float mu, float sd) {
  const float ac = fminf(fmaxf(act, -THRESH), THRESH);
  if (ac <= -THRESH) return log_ndtrf_((-INV_THRESH - mu) / sd) - LOG_EPS;
  if (ac >= THRESH) return log_ndtrf_(-(INV_THRESH - mu) / sd) - LOG_EPS;
  const float x = atanhf(ac);
  const float z = (x - mu) / sd;
  return -0.5f * z * z - logf(sd) - HALF_LOG_2PI - tanh_fldj(x);
}

__global__ void __launch_bounds__(256) policy_head_kernel(HeadArgs a) {
  float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};        // eval: sum loss, sum entropy, sum(l2 > l1), sum |ratio - 1|
  const int rows_pad = (a.rows + 255) & ~255;
  for (int row_raw = blockIdx.x * blockDim.x + threadIdx.x; row_raw < rows_pad; row_raw += gridDim.x * blockDim.x) {
  const bool valid = row_raw < a.rows;
  const int row = valid ? row_raw : a.rows - 1;  // invalid lanes recompute the last row and discard it (keeps waves whole)
  const float4 m = reinterpret_cast<const float4*>(a.ms)[row];
  const float mu[2] = {m.x, m.y};
  const float st[2] = {m.z, m.w};
  float sd[2];
  for (int d = 0; d < 2; ++d) sd[d] = softplusf_(st[d] + STD_INIT_INV) + STD_MIN;
  if (a.mode == 1) {  // PPOPolicy.get_action: tanh(mean)
    if (valid) {
      a.action[row * 2] = tanhf(mu[0]);
      a.action[row * 2 + 1] = tanhf(mu[1]);
    }
    continue;
  }
  float act[2];
  if (a.mode == 0) {  // sample_action: tanh(mean + std * eps); log_prob recomputes atanh(clip(a)) like the reference
    for (int d = 0; d < 2; ++d) act[d] = tanhf(mu[d] + sd[d] * a.eps[row * 2 + d]);
    if (valid) {
      a.action[row * 2] = act[0];
      a.action[row * 2 + 1] = act[1];
      a.log_pi[row] = tn_log_prob_dim(act[0], mu[0], sd[0]) + tn_log_prob_dim(act[1], mu[1], sd[1]);
    }
    continue;
  }
  // eval_action
  act[0] = a.action_in[row * 2];
  act[1] = a.action_in[row * 2 + 1];
  const int ag = row % a.n_agents;
  float lp = 0.0f, ent = 0.0f;
  float dlp_dmu[2], dlp_dsd[2], dent_dmu[2], dent_dsd[2];
  for (int d = 0; d < 2; ++d) {
    const float ac = fminf(fmaxf(act[d], -THRESH), THRESH);
    if (ac <= -THRESH) {
      const float u = (-INV_THRESH - mu[d]) / sd[d];
      lp += log_ndtrf_(u) - LOG_EPS;
      const float r = dlog_ndtrf_(u);
      dlp_dmu[d] = -r / sd[d];
      dlp_dsd[d] = -r * u / sd[d];
    } else if (ac >= THRESH) {
      const float u = (mu[d] - INV_THRESH) / sd[d];
      lp += log_ndtrf_(u) - LOG_EPS;
      const float r = dlog_ndtrf_(u);
      dlp_dmu[d] = r / sd[d];
      dlp_dsd[d] = -r * u / sd[d];
    } else {
      const float x = atanhf(ac);
      const float z = (x - mu[d]) / sd[d];
      lp += -0.5f * z * z - logf(sd[d]) - HALF_LOG_2PI - tanh_fldj(x);
      dlp_dmu[d] = z / sd[d];
      dlp_dsd[d] = (z * z - 1.0f) / sd[d];
    }
    const float eh = a.eps[ag * 2 + d];
    const float y = mu[d] + sd[d] * eh;
    ent += 0.5f + HALF_LOG_2PI + logf(sd[d]) + tanh_fldj(y);
    const float th = tanhf(y);
    dent_dmu[d] = -2.0f * th;
    dent_dsd[d] = 1.0f / sd[d] - 2.0f * th * eh;
  }
  if (valid) {
    a.log_pi[row] = lp;
    a.entropy[row] = ent;
  }
  if (a.dms == nullptr) continue;
  // PPO clipped surrogate (informarl.py:428-435):  mean(max(-rho A, -clip(rho) A)) - coef_ent * mean(H)
  const float A = a.adv[row];
  const float rho = expf(lp - a.log_pi_old[row]);
  const float l1 = -rho * A;
  const float l2 = -fminf(fmaxf(rho, 1.0f - a.clip_eps), 1.0f + a.clip_eps) * A;
  const float lsel = fmaxf(l1, l2);
  const float dl_dlp = ((l2 > l1) ? 0.0f : -rho * A) * a.inv_count;
  const float dl_dent = -a.coef_ent * a.inv_count;
  if (valid) {
    float4 o;
    const float dsd0 = dl_dlp * dlp_dsd[0] + dl_dent * dent_dsd[0];
    const float dsd1 = dl_dlp * dlp_dsd[1] + dl_dent * dent_dsd[1];
    o.x = dl_dlp * dlp_dmu[0] + dl_dent * dent_dmu[0];
    o.y = dl_dlp * dlp_dmu[1] + dl_dent * dent_dmu[1];
    o.z = dsd0 * sigmoidf_(st[0] + STD_INIT_INV);  // d softplus
    o.w = dsd1 * sigmoidf_(st[1] + STD_INIT_INV);
    reinterpret_cast<float4*>(a.dms)[row] = o;
  }
  // diagnostics: per-thread partial sums, reduced once per workgroup below
  if (valid) {
    acc[0] += lsel; acc[1] += ent; acc[2] += (l2 > l1) ? 1.0f : 0.0f; acc[3] += fabsf(rho - 1.0f);
  }
  }
  if (a.mode == 2 && a.dms != nullptr) block_atomic_sums<4>(acc, a.stats);
}

extern "C" int32_t dgppo_policy_head(const float* ms, const float* eps, const float* action_in, float* action,
                                     float* log_pi, float* entropy, int32_t rows, int32_t n_agents, int32_t mode,
                                     const float* log_pi_old, const float* adv, float* dms, float* stats, float clip_eps,
                                     float coef_ent, void* stream) {
  DGPPO_REQUIRE(rows >= 0 && n_agents >= 1 && mode >= 0 && mode <= 2, "policy_head: bad arguments");
  if (rows == 0) return 0;
  DGPPO_REQUIRE(ms, "policy_head: ms is NULL");
  DGPPO_REQUIRE(((uintptr_t)ms & 15) == 0, "policy_head: ms must be 16-byte aligned");
  if (mode == 0) DGPPO_REQUIRE(eps && action && log_pi, "policy_head(sample): NULL operand");
  if (mode == 1) DGPPO_REQUIRE(action, "policy_head(mode): NULL operand");
  if (mode == 2) {
    DGPPO_REQUIRE(eps && action_in && log_pi && entropy, "policy_head(eval): NULL operand");
    if (dms) DGPPO_REQUIRE(log_pi_old && adv && stats && ((uintptr_t)dms & 15) == 0, "policy_head(eval+loss): NULL operand");
  }
  HeadArgs a{};
  a.ms = ms; a.eps = eps; a.action_in = action_in; a.action = action; a.log_pi = log_pi; a.entropy = entropy;
  a.rows = rows; a.n_agents = n_agents; a.mode = mode; a.log_pi_old = log_pi_old; a.adv = adv; a.dms = dms; a.stats = stats;
  a.clip_eps = clip_eps; a.coef_ent = coef_ent; a.inv_count = 1.0f / (float)rows;
  const int blocks = cdiv(rows, 256);
  hipLaunchKernelGGL(policy_head_kernel, dim3(blocks < 512 ? blocks : 512), dim3(256), 0, (hipStream_t)stream, a);
  DGPPO_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// value loss 1/2 (v - target)^2 mean (optax.l2_loss().mean()), gradient and sum
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) value_loss_kernel(const float* __restrict__ v, const float* __restrict__ target,
                                                         float* __restrict__ dv, float* __restrict__ stats, int count,
                                                         float inv_count) {
  float l[1] = {0.0f};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
    const float d = v[i] - target[i];
    dv[i] = d * inv_count;
    l[0] += 0.5f * d * d;
  }
  block_atomic_sums<1>(l, stats);
}

extern "C" int32_t dgppo_value_loss(const float* v, const float* target, float* dv, float* stats, int32_t count,
                                    void* stream) {
  DGPPO_REQUIRE(count >= 0, "value_loss: count < 0");
  if (count == 0) return 0;
  DGPPO_REQUIRE(v && target && dv && stats, "value_loss: NULL operand");
  const int blocks = cdiv(count, 256);
  hipLaunchKernelGGL(value_loss_kernel, dim3(blocks < 256 ? blocks : 256), dim3(256), 0, (hipStream_t)stream, v, target, dv,
                     stats, count, 1.0f / (float)count);
  DGPPO_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// mean over the agents of a graph (value.py:33) and its backward; relu backward; small utilities
// ---------------------------------------------------------------------------------------------------------------------
__global__ void mean_agents_kernel(const float* __restrict__ x, float* __restrict__ y, int G, int n, int D, int backward,
                                   const float* __restrict__ relu_mask) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (backward) {  // x = dP [G, D] -> y = dX [G, n, D]  (optionally through the ReLU whose output is relu_mask)
    if (idx >= G * n * D) return;
    const int d = idx % D, g = idx / (n * D);
    // backward == 1: y = x / n (gradient of the mean);  backward == 2: y += x (broadcast-add of a per-graph row: the
    // tiled global feature of DecRStateFn(use_global_info=True), value.py:66-68, whose mean factor is already in x)
    const float v = (backward == 2) ? y[idx] + x[(size_t)g * D + d] : x[(size_t)g * D + d] / (float)n;
    y[idx] = (relu_mask == nullptr || relu_mask[idx] > 0.0f) ? v : 0.0f;
  } else {
    if (idx >= G * D) return;
    const int d = idx % D, g = idx / D;
    float acc = 0.0f;
    for (int i = 0; i < n; ++i) acc += x[((size_t)g * n + i) * D + d];
    y[idx] = acc / (float)n;
  }
}

extern "C" int32_t dgppo_mean_agents(const float* x, float* y, int32_t G, int32_t n, int32_t D, int32_t backward,
                                     const float* relu_mask, void* stream) {
  DGPPO_REQUIRE(G >= 0 && n >= 1 && D >= 1, "mean_agents: bad sizes");
  DGPPO_REQUIRE(backward >= 0 && backward <= 2, "mean_agents: backward must be 0, 1 or 2");
  if (G == 0) return 0;
  DGPPO_REQUIRE(x && y, "mean_agents: NULL operand");
  const long total = backward ? (long)G * n * D : (long)G * D;
  hipLaunchKernelGGL(mean_agents_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, y, G, n, D, backward,
                     backward ? relu_mask : nullptr);
  DGPPO_LAUNCH_CHECK();
  return 0;
}

// dx = dy * (y > 0), in place on dy
__global__ void relu_bwd_kernel(float* __restrict__ dy, const float* __restrict__ y, long count) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count && !(y[i] > 0.0f)) dy[i] = 0.0f;
}

extern "C" int32_t dgppo_relu_bwd(float* dy, const float* y, int64_t count, void* stream) {
  DGPPO_REQUIRE(count >= 0, "relu_bwd: count < 0");
  if (count == 0) return 0;
  DGPPO_REQUIRE(dy && y, "relu_bwd: NULL operand");
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(cdiv(count, 256)), dim3(256), 0, (hipStream_t)stream, dy, y, (long)count);
  DGPPO_LAUNCH_CHECK();
  return 0;
}
