// Dense (fully connected) building blocks on the fp32 matrix cores of gfx950 (v_mfma_f32_16x16x4_f32):
//   dgppo_dense_fwd   Y[M,N]  = act(X[M,K] * W[K,N] + b)        (also  dX = dY * W^T via trans_w)
//   dgppo_dense_bwd_w dW[K,N] += X[M,K]^T * dY[M,N],  db[N] += colsum(dY)
// They replace flax nn.Dense as used by dgppo/nn/mlp.py:19-22, dgppo/nn/gnn.py:86-110, dgppo/nn/rnn.py:19-21 (GRUCell
// input/recurrent projections), dgppo/algo/module/policy.py:67-70 and dgppo/algo/module/value.py:41,76 — and the
// jax.grad of those (dgppo/algo/informarl.py:377,440; dgppo/algo/dgppo.py:316).
// fp32-input MFMA is bit-for-bit a k-ordered fmaf chain (exact fp32, no reduced precision).
#include "common.h"

using f32x4 = __attribute__((ext_vector_type(4))) float;

#define DENSE_ROWS 64  // rows per workgroup (4 waves x 16)

struct DenseArgs {
  const float* X; int ldx;
  const float* W; int ldw;
  const float* bias;
  float* Y; int ldy;
  int M, K, N;
  int act;         // 0 none, 1 relu
  int accumulate;  // Y += result
  int trans_w;     // use W^T: result[m,n] = sum_k X[m,k] * W[n,k]
};

// One workgroup = 64 rows x all N columns.  X tile staged in LDS (coalesced), W fragments straight from L1/L2.
template <int NT>
__global__ void __launch_bounds__(256) dense_fwd_kernel(DenseArgs a) {
  extern __shared__ float xs[];  // [64][K+1]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row0 = blockIdx.x * DENSE_ROWS;
  const int K = a.K, N = a.N, Kl = K + 1;
  for (int idx = tid; idx < DENSE_ROWS * K; idx += 256) {
    const int r = idx / K, k = idx - r * K;
    const int row = row0 + r;
    xs[r * Kl + k] = (row < a.M) ? a.X[(size_t)row * a.ldx + k] : 0.0f;
  }
  __syncthreads();
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int li = lane & 15, lq = lane >> 4;
  const float* xrow = xs + (wave * 16 + li) * Kl;
  for (int k0 = 0; k0 < K; k0 += 4) {
    const int k = k0 + lq;
    const float av = (k < K) ? xrow[k] : 0.0f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int col = t * 16 + li;
      float bv = 0.0f;
      if (k < K && col < N) bv = a.trans_w ? a.W[(size_t)col * a.ldw + k] : a.W[(size_t)k * a.ldw + col];
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[t], 0, 0, 0);
    }
  }
  // C/D layout: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = t * 16 + li;
    if (col >= N) continue;
    const float bb = a.bias ? a.bias[col] : 0.0f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = row0 + wave * 16 + lq * 4 + r;
      if (row >= a.M) continue;
      float v = acc[t][r] + bb;
      float* dst = a.Y + (size_t)row * a.ldy + col;
      if (a.accumulate) v += *dst;
      if (a.act == 1) v = fmaxf(v, 0.0f);
      *dst = v;
    }
  }
}

template <int NT>
static void launch_dense(const DenseArgs& a, hipStream_t s) {
  const size_t smem = (size_t)DENSE_ROWS * (a.K + 1) * sizeof(float);
  hipLaunchKernelGGL(dense_fwd_kernel<NT>, dim3(cdiv(a.M, DENSE_ROWS)), dim3(256), smem, s, a);
}

int32_t dense_fwd_launch(const DenseArgs& a, hipStream_t s) {
  DGPPO_REQUIRE(a.M >= 0 && a.K >= 1 && a.N >= 1, "dense: bad shape M=%d K=%d N=%d", a.M, a.K, a.N);
  DGPPO_REQUIRE(a.N <= 192 && a.K <= 256, "dense: N <= 192 and K <= 256 supported (N=%d K=%d)", a.N, a.K);
  DGPPO_REQUIRE(a.X && a.W && a.Y, "dense: NULL operand");
  DGPPO_REQUIRE(a.ldx >= a.K && a.ldy >= a.N, "dense: leading dimensions too small");
  if (a.M == 0) return 0;
  const int nt = cdiv(a.N, 16);
  if (nt <= 1) launch_dense<1>(a, s);
  else if (nt <= 2) launch_dense<2>(a, s);
  else if (nt <= 4) launch_dense<4>(a, s);
  else if (nt <= 6) launch_dense<6>(a, s);
  else launch_dense<12>(a, s);
  DGPPO_LAUNCH_CHECK();
  return 0;
}

// ---- weight gradient: dW[K,N] += X^T dY, db[N] += colsum(dY) -------------------------------------------------------
struct DenseBwdWArgs {
  const float* X; int ldx;
  const float* dY; int ldy;
  float* dW; int ldw;
  float* db;  // may be NULL
  int M, K, N;
  int rows_per_block;
};

// wave w owns output row-tiles kt = w, w+4, ... (KTW of them) x all NT column tiles; loops over the block's rows in
// steps of 4 (the MFMA k dimension); fragments come straight from global (each 4-row slab is shared via L1).
template <int NT, int KTW>
__global__ void __launch_bounds__(256) dense_bwd_w_kernel(DenseBwdWArgs a) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lq = lane >> 4;
  const int m_begin = blockIdx.x * a.rows_per_block;
  const int m_end = min(a.M, m_begin + a.rows_per_block);
  f32x4 acc[KTW][NT];
#pragma unroll
  for (int i = 0; i < KTW; ++i)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float colsum[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) colsum[t] = 0.0f;
  const int KT = (a.K + 15) / 16;
  for (int m0 = m_begin; m0 < m_end; m0 += 4) {
    const int m = m0 + lq;
    const bool mv = m < m_end;
    float bv[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int col = t * 16 + li;
      bv[t] = (mv && col < a.N) ? a.dY[(size_t)m * a.ldy + col] : 0.0f;
      colsum[t] += bv[t];
    }
#pragma unroll
    for (int i = 0; i < KTW; ++i) {
      const int kt = wave + 4 * i;
      if (kt >= KT) continue;
      const int kcol = kt * 16 + li;
      const float av = (mv && kcol < a.K) ? a.X[(size_t)m * a.ldx + kcol] : 0.0f;
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[t], acc[i][t], 0, 0, 0);
    }
  }
#pragma unroll
  for (int i = 0; i < KTW; ++i) {
    const int kt = wave + 4 * i;
    if (kt >= KT) continue;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int col = t * 16 + li;
      if (col >= a.N) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int krow = kt * 16 + lq * 4 + r;
        if (krow < a.K) atomicAdd(a.dW + (size_t)krow * a.ldw + col, acc[i][t][r]);
      }
    }
  }
  if (a.db != nullptr && wave == 0) {
    // lanes with equal (lane & 15) hold partial sums of the same column (lq = 0..3)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float v = colsum[t];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      const int col = t * 16 + li;
      if (lq == 0 && col < a.N) atomicAdd(a.db + col, v);
    }
  }
}

template <int NT>
static void launch_bwd_w(const DenseBwdWArgs& a, int grid, hipStream_t s) {
  const int ktw = cdiv(cdiv(a.K, 16), 4);
  if (ktw <= 1) hipLaunchKernelGGL((dense_bwd_w_kernel<NT, 1>), dim3(grid), dim3(256), 0, s, a);
  else if (ktw == 2) hipLaunchKernelGGL((dense_bwd_w_kernel<NT, 2>), dim3(grid), dim3(256), 0, s, a);
  else if (ktw == 3) hipLaunchKernelGGL((dense_bwd_w_kernel<NT, 3>), dim3(grid), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((dense_bwd_w_kernel<NT, 4>), dim3(grid), dim3(256), 0, s, a);
}

int32_t dense_bwd_w_launch(DenseBwdWArgs a, hipStream_t s) {
  DGPPO_REQUIRE(a.M >= 0 && a.K >= 1 && a.N >= 1, "dense_bwd_w: bad shape");
  DGPPO_REQUIRE(a.N <= 192 && a.K <= 256, "dense_bwd_w: N <= 192 and K <= 256 supported (N=%d K=%d)", a.N, a.K);
  DGPPO_REQUIRE(a.X && a.dY && a.dW, "dense_bwd_w: NULL operand");
  if (a.M == 0) return 0;
  // ~1024 workgroups, rows per block a multiple of 4
  int rpb = cdiv(a.M, 1024);
  rpb = ((rpb + 3) / 4) * 4;
  if (rpb < 64) rpb = 64;
  a.rows_per_block = rpb;
  const int grid = cdiv(a.M, rpb);
  const int nt = cdiv(a.N, 16);
  if (nt <= 1) launch_bwd_w<1>(a, grid, s);
  else if (nt <= 2) launch_bwd_w<2>(a, grid, s);
  else if (nt <= 4) launch_bwd_w<4>(a, grid, s);
  else if (nt <= 6) launch_bwd_w<6>(a, grid, s);
  else launch_bwd_w<12>(a, grid, s);
  DGPPO_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t dgppo_dense_fwd(const float* X, int32_t ldx, const float* W, int32_t ldw, const float* bias, float* Y,
                                   int32_t ldy, int32_t M, int32_t K, int32_t N, int32_t act, int32_t accumulate,
                                   int32_t trans_w, void* stream) {
  DenseArgs a{X, ldx, W, ldw, bias, Y, ldy, M, K, N, act, accumulate, trans_w};
  return dense_fwd_launch(a, (hipStream_t)stream);
}

extern "C" int32_t dgppo_dense_bwd_w(const float* X, int32_t ldx, const float* dY, int32_t ldy, float* dW, int32_t ldw,
                                     float* db, int32_t M, int32_t K, int32_t N, void* stream) {
  DenseBwdWArgs a{X, ldx, dY, ldy, dW, ldw, db, M, K, N, 0};
  return dense_bwd_w_launch(a, (hipStream_t)stream);
}
