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

// One workgroup = 64 rows x all N columns.  The X tile is staged once in LDS (coalesced, float4 when aligned).  The
// N/16 column tiles are dealt to CG = min(4, tiles) column groups (one or more waves each), the 4 row tiles of 16 rows to
// the remaining 4/CG row groups.  K is walked in chunks of 64: a wave first pulls its W fragments of the chunk into
// registers (16 x NTW values per lane, read once per workgroup from L1/L2), then runs every row tile it owns against
// them, so the inner loop issues one LDS read (the A fragment) per NTW MFMAs and no global loads.
template <int NTW, int RTW>
__global__ void __launch_bounds__(256) dense_fwd_kernel(DenseArgs a) {
  extern __shared__ float xs[];  // [64][Kl]
  constexpr int CG = (RTW == 4) ? 4 : ((RTW == 2) ? 2 : 1);   // column groups
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row0 = blockIdx.x * DENSE_ROWS;
  const int K = a.K, N = a.N, Kl = K | 1;                      // odd row stride: conflict-free A-fragment reads
  const bool vec = ((K & 3) == 0) && ((a.ldx & 3) == 0) && ((reinterpret_cast<uintptr_t>(a.X) & 15) == 0);
  if (vec) {
    const int K4 = K >> 2;
    for (int idx = tid; idx < DENSE_ROWS * K4; idx += 256) {
      const int r = idx / K4, q = idx - r * K4;
      const int row = row0 + r;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < a.M) v = *reinterpret_cast<const float4*>(a.X + (size_t)row * a.ldx + 4 * q);
      float* d = xs + r * Kl + 4 * q;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
  } else {
    for (int idx = tid; idx < DENSE_ROWS * K; idx += 256) {
      const int r = idx / K, k = idx - r * K;
      const int row = row0 + r;
      xs[r * Kl + k] = (row < a.M) ? a.X[(size_t)row * a.ldx + k] : 0.0f;
    }
  }
  __syncthreads();
  const int li = lane & 15, lq = lane >> 4;
  const int cg = wave % CG, rg = wave / CG;
  f32x4 acc[RTW][NTW];
#pragma unroll
  for (int r = 0; r < RTW; ++r)
#pragma unroll
    for (int t = 0; t < NTW; ++t) acc[r][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int kc = 0; kc < K; kc += 64) {
    float breg[16][NTW];
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      const int k = kc + kk * 4 + lq;
#pragma unroll
      for (int t = 0; t < NTW; ++t) {
        const int col = (cg + CG * t) * 16 + li;
        float bv = 0.0f;
        if (k < K && col < N) bv = a.trans_w ? a.W[(size_t)col * a.ldw + k] : a.W[(size_t)k * a.ldw + col];
        breg[kk][t] = bv;
      }
    }
#pragma unroll
    for (int r = 0; r < RTW; ++r) {
      const float* xrow = xs + ((rg * RTW + r) * 16 + li) * Kl + kc + lq;
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) {
        const float av = (kc + kk * 4 + lq < K) ? xrow[kk * 4] : 0.0f;
#pragma unroll
        for (int t = 0; t < NTW; ++t) acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, breg[kk][t], acc[r][t], 0, 0, 0);
      }
    }
  }
  // C/D layout: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    const int col = (cg + CG * t) * 16 + li;
    if (col >= N) continue;
    const float bb = a.bias ? a.bias[col] : 0.0f;
#pragma unroll
    for (int r = 0; r < RTW; ++r) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = row0 + (rg * RTW + r) * 16 + lq * 4 + j;
        if (row >= a.M) continue;
        float v = acc[r][t][j] + bb;
        float* dst = a.Y + (size_t)row * a.ldy + col;
        if (a.accumulate) v += *dst;
        if (a.act == 1) v = fmaxf(v, 0.0f);
        *dst = v;
      }
    }
  }
}

template <int NTW, int RTW>
static void launch_dense(const DenseArgs& a, hipStream_t s) {
  const size_t smem = (size_t)DENSE_ROWS * (a.K | 1) * sizeof(float);
  hipLaunchKernelGGL((dense_fwd_kernel<NTW, RTW>), dim3(cdiv(a.M, DENSE_ROWS)), dim3(256), smem, s, a);
}

int32_t dense_fwd_launch(const DenseArgs& a, hipStream_t s) {
  DGPPO_REQUIRE(a.M >= 0 && a.K >= 1 && a.N >= 1, "dense: bad shape M=%d K=%d N=%d", a.M, a.K, a.N);
  DGPPO_REQUIRE(a.N <= 192 && a.K <= 256, "dense: N <= 192 and K <= 256 supported (N=%d K=%d)", a.N, a.K);
  DGPPO_REQUIRE(a.X && a.W && a.Y, "dense: NULL operand");
  DGPPO_REQUIRE(a.ldx >= a.K && a.ldy >= a.N, "dense: leading dimensions too small");
  if (a.M == 0) return 0;
  const int nt = cdiv(a.N, 16);
  if (nt >= 9) launch_dense<3, 4>(a, s);        // N in (128, 192]
  else if (nt >= 5) launch_dense<2, 4>(a, s);   // N in (64, 128]
  else if (nt == 4) launch_dense<1, 4>(a, s);   // N in (48, 64]
  else if (nt == 3) launch_dense<2, 2>(a, s);   // N in (32, 48]
  else if (nt == 2) launch_dense<1, 2>(a, s);   // N in (16, 32]
  else launch_dense<1, 1>(a, s);                // N <= 16
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

// A workgroup walks its chunk of rows in tiles of 32: X[32,K] and dY[32,N] are staged in LDS with coalesced (float4)
// loads, then wave w accumulates the output row-tiles kt = w, w+4, ... (KTW of them) x all NT column tiles with MFMA over
// the tile's 8 four-row slabs (A[i=k][kk=m] = X[m][k], B[kk=m][j] = dY[m][j]); column sums for db come from the same
// LDS tile.  One fp32 atomicAdd per output element per workgroup at the end.
#define BW_ROWS 32
template <int NT, int KTW>
__global__ void __launch_bounds__(256) dense_bwd_w_kernel(DenseBwdWArgs a) {
  extern __shared__ float sm[];
  const int K = a.K, N = a.N;
  const int Kl = (K + 3) / 4 * 4 + 4, Nl = (N + 3) / 4 * 4 + 4;   // row strides (multiples of 4 floats, not of 32)
  float* xs = sm;                  // [32][Kl]
  float* ys = sm + BW_ROWS * Kl;   // [32][Nl]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lq = lane >> 4;
  const int m_begin = blockIdx.x * a.rows_per_block;
  const int m_end = min(a.M, m_begin + a.rows_per_block);
  f32x4 acc[KTW][NT];
#pragma unroll
  for (int i = 0; i < KTW; ++i)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float colsum = 0.0f;
  const int KT = (K + 15) / 16;
  const bool vx = ((K & 3) == 0) && ((a.ldx & 3) == 0) && ((reinterpret_cast<uintptr_t>(a.X) & 15) == 0);
  const bool vy = ((N & 3) == 0) && ((a.ldy & 3) == 0) && ((reinterpret_cast<uintptr_t>(a.dY) & 15) == 0);
  for (int m0 = m_begin; m0 < m_end; m0 += BW_ROWS) {
    if (vx) {
      const int K4 = K >> 2;
      for (int idx = tid; idx < BW_ROWS * K4; idx += 256) {
        const int r = idx / K4, q = idx - r * K4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (m0 + r < m_end) v = *reinterpret_cast<const float4*>(a.X + (size_t)(m0 + r) * a.ldx + 4 * q);
        *reinterpret_cast<float4*>(xs + r * Kl + 4 * q) = v;
      }
    } else {
      for (int idx = tid; idx < BW_ROWS * K; idx += 256) {
        const int r = idx / K, k = idx - r * K;
        xs[r * Kl + k] = (m0 + r < m_end) ? a.X[(size_t)(m0 + r) * a.ldx + k] : 0.0f;
      }
    }
    if (vy) {
      const int N4 = N >> 2;
      for (int idx = tid; idx < BW_ROWS * N4; idx += 256) {
        const int r = idx / N4, q = idx - r * N4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (m0 + r < m_end) v = *reinterpret_cast<const float4*>(a.dY + (size_t)(m0 + r) * a.ldy + 4 * q);
        *reinterpret_cast<float4*>(ys + r * Nl + 4 * q) = v;
      }
    } else {
      for (int idx = tid; idx < BW_ROWS * N; idx += 256) {
        const int r = idx / N, c = idx - r * N;
        ys[r * Nl + c] = (m0 + r < m_end) ? a.dY[(size_t)(m0 + r) * a.ldy + c] : 0.0f;
      }
    }
    __syncthreads();
    if (a.db != nullptr && tid < N) {
      float cs = 0.0f;
#pragma unroll 8
      for (int r = 0; r < BW_ROWS; ++r) cs += ys[r * Nl + tid];
      colsum += cs;
    }
#pragma unroll
    for (int s4 = 0; s4 < BW_ROWS / 4; ++s4) {
      const int r = s4 * 4 + lq;
      float bv[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) bv[t] = (t * 16 + li < N) ? ys[r * Nl + t * 16 + li] : 0.0f;
#pragma unroll
      for (int i = 0; i < KTW; ++i) {
        const int kt = wave + 4 * i;
        if (kt >= KT) continue;
        const float av = (kt * 16 + li < K) ? xs[r * Kl + kt * 16 + li] : 0.0f;
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[t], acc[i][t], 0, 0, 0);
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < KTW; ++i) {
    const int kt = wave + 4 * i;
    if (kt >= KT) continue;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int col = t * 16 + li;
      if (col >= N) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int krow = kt * 16 + lq * 4 + r;
        if (krow < K) atomicAdd(a.dW + (size_t)krow * a.ldw + col, acc[i][t][r]);
      }
    }
  }
  if (a.db != nullptr && tid < N) atomicAdd(a.db + tid, colsum);
}

template <int NT>
static void launch_bwd_w(const DenseBwdWArgs& a, int grid, hipStream_t s) {
  const int ktw = cdiv(cdiv(a.K, 16), 4);
  const size_t smem = sizeof(float) * BW_ROWS * (((a.K + 3) / 4 * 4 + 4) + ((a.N + 3) / 4 * 4 + 4));
  if (ktw <= 1) hipLaunchKernelGGL((dense_bwd_w_kernel<NT, 1>), dim3(grid), dim3(256), smem, s, a);
  else if (ktw == 2) hipLaunchKernelGGL((dense_bwd_w_kernel<NT, 2>), dim3(grid), dim3(256), smem, s, a);
  else if (ktw == 3) hipLaunchKernelGGL((dense_bwd_w_kernel<NT, 3>), dim3(grid), dim3(256), smem, s, a);
  else hipLaunchKernelGGL((dense_bwd_w_kernel<NT, 4>), dim3(grid), dim3(256), smem, s, a);
}

int32_t dense_bwd_w_launch(DenseBwdWArgs a, hipStream_t s) {
  DGPPO_REQUIRE(a.M >= 0 && a.K >= 1 && a.N >= 1, "dense_bwd_w: bad shape");
  DGPPO_REQUIRE(a.N <= 192 && a.K <= 256, "dense_bwd_w: N <= 192 and K <= 256 supported (N=%d K=%d)", a.N, a.K);
  DGPPO_REQUIRE(a.X && a.dY && a.dW, "dense_bwd_w: NULL operand");
  if (a.M == 0) return 0;
  // ~1024 workgroups, rows per block a multiple of the 32-row tile
  int rpb = cdiv(a.M, 1024);
  rpb = ((rpb + BW_ROWS - 1) / BW_ROWS) * BW_ROWS;
  if (rpb < 2 * BW_ROWS) rpb = 2 * BW_ROWS;
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
