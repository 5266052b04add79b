// Fused row pipelines (fp32 MFMA): chains of row-local layers that the reference runs back to back on the same rows,
// executed by one persistent kernel so that the activations between them never leave the CU.
//
// Reference arithmetic replaced (file:line relative to /root/reference):
//   MLP block (Dense -> LayerNorm -> ReLU) x 2     dgppo/nn/mlp.py:17-29   (hid_sizes (64, 64), act_final=True)
//   GRUCell input projection  x W_i + b_i          dgppo/nn/rnn.py:14-30   (flax GRUCell: dense_i of the r|z|n gates)
//   as composed by                                  dgppo/algo/module/policy.py:191-212, value.py:58-80
#include "common.h"

using f32x4 = __attribute__((ext_vector_type(4))) float;
#define FZ_H 64            // hidden width of every layer in the chain
#define FZ_HL 66           // LDS row stride = 2 (mod 32): A-fragment reads touch banks (2 li + lq) mod 32, distinct in each half-wave (65 was 2-way: r03_nn_counters.json)
#define FZ_RB 32           // rows per tile (2 row tiles of 16 per wave)
#define FZ_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// sum over the 16 lanes of a DPP row (= the 16 columns a wave owns), result valid in every lane.  DPP moves stay inside
// the VALU; __shfl_xor would go through the LDS crossbar (ds_bpermute) with a round trip per step.
__device__ inline float row16_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, false));   // row_half_mirror
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, false));   // row_mirror
  return v;
}

struct MlpGiArgs {
  const float* X; int ldx; int M;
  const float *W1, *b1, *g1, *be1, *W2, *b2, *g2, *be2, *Wi, *bi;
  float *p1, *y1, *st1, *p2, *y2, *st2;   // saved for the backward (all NULL in inference)
  float* gi;                              // [M, 192]
};

// One workgroup = 4 waves; wave w owns hidden columns 16w..16w+15 of the 64-wide layers and column tiles w, w+4, w+8 of
// the 192-wide gate projection.  All weight fragments of the wave (16 + 16 + 48 k-step registers) are loaded once and
// stay resident while the workgroup walks 32-row tiles; the next tile's rows are requested while the current one
// computes.  LayerNorm statistics: 16-lane shuffle reduction inside the wave, 4-way exchange through LDS.  Four LDS-only
// barriers per tile.
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 8))) mlp_gi_fwd_kernel(MlpGiArgs a) {
  extern __shared__ float sm[];
  float* s_x = sm;                              // [RB][HL]  input rows
  float* s_y1 = s_x + FZ_RB * FZ_HL;            // [RB][HL]  relu(LN(x W1 + b1))
  float* s_y2 = s_y1 + FZ_RB * FZ_HL;           // [RB][HL]  relu(LN(y1 W2 + b2))
  float* s_red = s_y2 + FZ_RB * FZ_HL;          // [RB][8]   per-row (sum, sum of squares) of each of the 4 waves
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 15, lq = lane >> 4;
  const int c = w * 16 + li;
  float w1[16], w2[16], wi[16][3];
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) {
    const int k = kk * 4 + lq;
    w1[kk] = a.W1[k * FZ_H + c];
    w2[kk] = a.W2[k * FZ_H + c];
#pragma unroll
    for (int t = 0; t < 3; ++t) wi[kk][t] = a.Wi[k * 192 + t * 64 + c];
  }
  const float b1 = a.b1[c], g1 = a.g1[c], e1 = a.be1[c], b2 = a.b2[c], g2 = a.g2[c], e2 = a.be2[c];
  const float bi0 = a.bi[c], bi1 = a.bi[64 + c], bi2 = a.bi[128 + c];
  const int n_tiles = (a.M + FZ_RB - 1) / FZ_RB;
  const bool vec = ((a.ldx & 3) == 0) && ((reinterpret_cast<uintptr_t>(a.X) & 15) == 0);
  float xpf[2][4];
  auto fetch = [&](int tile) {            // 32 rows x 16 float4 = 512 slots, 2 per lane; rows clamped
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int idx = u * 256 + tid, r = idx >> 4, q = idx & 15;
      int row = tile * FZ_RB + r;
      row = row < a.M ? row : a.M - 1;
      const float* p = a.X + (size_t)row * a.ldx + 4 * q;
      if (vec) { const float4 v = *reinterpret_cast<const float4*>(p); xpf[u][0] = v.x; xpf[u][1] = v.y; xpf[u][2] = v.z; xpf[u][3] = v.w; }
      else { xpf[u][0] = p[0]; xpf[u][1] = p[1]; xpf[u][2] = p[2]; xpf[u][3] = p[3]; }
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int idx = u * 256 + tid, r = idx >> 4, q = idx & 15;
      float* d = s_x + r * FZ_HL + 4 * q;
      d[0] = xpf[u][0]; d[1] = xpf[u][1]; d[2] = xpf[u][2]; d[3] = xpf[u][3];
    }
  };
  // y = relu(LN(v)) for this lane's 8 elements (2 row tiles x 4 rows, column c); v is overwritten
  auto ln_relu = [&](float (&v)[2][4], float g, float e, float* s_out, float* y_out, float* st_out, int row0) {
    // partial sums over the wave's 16 columns
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float s = row16_sum(v[rt][r]), q = row16_sum(v[rt][r] * v[rt][r]);
        if (li == 0) { float* d = s_red + (rt * 16 + lq * 4 + r) * 8 + 2 * w; d[0] = s; d[1] = q; }
      }
    FZ_LDS_BARRIER();
    float4 pa[2][4], pb[2][4];            // all 16 reads in one LDS round trip
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rl = rt * 16 + lq * 4 + r;
        pa[rt][r] = *reinterpret_cast<const float4*>(s_red + rl * 8);
        pb[rt][r] = *reinterpret_cast<const float4*>(s_red + rl * 8 + 4);
      }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rl = rt * 16 + lq * 4 + r;
        const float4 p0 = pa[rt][r], p1 = pb[rt][r];
        const float mean = ((p0.x + p0.z) + (p1.x + p1.z)) * (1.0f / 64.0f);
        const float mean2 = ((p0.y + p0.w) + (p1.y + p1.w)) * (1.0f / 64.0f);
        const float var = fmaxf(mean2 - mean * mean, 0.0f);
        const float rstd = rsqrtf(var + 1e-6f);
        const float y = fmaxf((v[rt][r] - mean) * rstd * g + e, 0.0f);
        s_out[rl * FZ_HL + c] = y;
        const int row = row0 + rl;
        if (y_out != nullptr && row < a.M) {
          y_out[(size_t)row * FZ_H + c] = y;
          if (c == 0) { st_out[(size_t)row * 2] = mean; st_out[(size_t)row * 2 + 1] = rstd; }
        }
      }
    FZ_LDS_BARRIER();
  };
  int tile = blockIdx.x;
  if (tile < n_tiles) { fetch(tile); commit(); }
  __syncthreads();
  for (; tile < n_tiles; tile += gridDim.x) {
    const int row0 = tile * FZ_RB;
    const int nxt = tile + gridDim.x;
    const bool more = nxt < n_tiles;
    if (more) fetch(nxt);
    __builtin_amdgcn_sched_barrier(0);
    float areg[2][16];
    f32x4 acc[2];
    float v[2][4];
    // ---- layer 1 ----
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) areg[rt][kk] = s_x[(rt * 16 + li) * FZ_HL + kk * 4 + lq];
    __builtin_amdgcn_sched_barrier(0);    // keep the 32 reads together: one LDS round trip, then MFMAs back to back
    acc[0] = acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 16; ++kk)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[rt][kk], w1[kk], acc[rt], 0, 0, 0);
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[rt][r] = acc[rt][r] + b1;
        const int row = row0 + rt * 16 + lq * 4 + r;
        if (a.p1 != nullptr && row < a.M) a.p1[(size_t)row * FZ_H + c] = v[rt][r];
      }
    // (the first barrier inside ln_relu also retires every read of s_x: the next tile's rows may be committed after it)
    ln_relu(v, g1, e1, s_y1, a.y1, a.st1, row0);
    if (more) commit();
    // ---- layer 2 ----
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) areg[rt][kk] = s_y1[(rt * 16 + li) * FZ_HL + kk * 4 + lq];
    __builtin_amdgcn_sched_barrier(0);
    acc[0] = acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 16; ++kk)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[rt][kk], w2[kk], acc[rt], 0, 0, 0);
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[rt][r] = acc[rt][r] + b2;
        const int row = row0 + rt * 16 + lq * 4 + r;
        if (a.p2 != nullptr && row < a.M) a.p2[(size_t)row * FZ_H + c] = v[rt][r];
      }
    ln_relu(v, g2, e2, s_y2, a.y2, a.st2, row0);
    // ---- gate projection ----
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) areg[rt][kk] = s_y2[(rt * 16 + li) * FZ_HL + kk * 4 + lq];
    __builtin_amdgcn_sched_barrier(0);
    f32x4 ag[2][3];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int t = 0; t < 3; ++t) ag[rt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 16; ++kk)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int t = 0; t < 3; ++t) ag[rt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[rt][kk], wi[kk][t], ag[rt][t], 0, 0, 0);
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + rt * 16 + lq * 4 + r;
        if (row < a.M) {
          float* o = a.gi + (size_t)row * 192 + c;
          o[0] = ag[rt][0][r] + bi0; o[64] = ag[rt][1][r] + bi1; o[128] = ag[rt][2][r] + bi2;
        }
      }
  }
}

// ---- the same chain, rows per WAVE (round 3) ----------------------------------------------------------------------------
// mlp_gi_fwd_kernel above splits the 64 columns of a 32-row tile over the 4 waves of a workgroup: every LayerNorm needs an
// exchange through LDS and two workgroup barriers, four barriers per tile at 2 workgroups per CU (232 VGPRs) — its matrix cores
// are busy 37 % of the time and a launch over 32 768 rows (a rollout step) takes 24 us for 8 us of MFMA work.  Here a WAVE owns
// 16 rows through the whole chain and all 64 / 192 output columns of every layer:
//   * the three weight matrices (80 KB) sit in LDS once per CU ([k][c] rows, stride = 1 (mod 64) x 16 so that the B fragment of
//     a k-step — lane (li, lq) reads W[lq * 16 + kk][ct * 16 + li] — touches 64 different banks); one workgroup of 16 waves
//     per CU shares them (gfx950: a workgroup may declare all 160 KB);
//   * LayerNorm statistics are sums over the lane's 4 column tiles + one 16-lane DPP reduction: no exchange, NO barrier after
//     the weights are staged;
//   * a layer's output (C/D layout) becomes the next layer's A operand through a wave-private 16 x 68 LDS tile (DS operations of
//     a wave execute in order); the contraction index is permuted so that a lane's 16 k-values are 16 consecutive columns
//     (four 16-byte reads; the input rows come straight from global memory the same way).
#define RW_WL 65
#define RW_WIL 193
#define RW_YL 68
#define RW_WAVES 16
__global__ void __launch_bounds__(64 * RW_WAVES) mlp_gi_fwd_rw_kernel(MlpGiArgs a) {
  extern __shared__ float sm[];
  float* sW1 = sm;                               // [64][65]
  float* sW2 = sW1 + 64 * RW_WL;                 // [64][65]
  float* sWi = sW2 + 64 * RW_WL;                 // [64][193]
  float* sP = sWi + 64 * RW_WIL;                 // b1 g1 be1 b2 g2 be2 (6 x 64), bi (192)
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 15, lq = lane >> 4;
  float* sY = sP + 576 + w * (16 * RW_YL);       // wave-private transposition tile
  {   // the 80 KB of weights: five 16-byte loads per thread, ALL requested before the first LDS write (a load -> store loop
      // is one trip to L2 per iteration)
    static_assert(RW_WAVES == 16, "the staging below assumes 1024 threads");
    const float4 v1 = reinterpret_cast<const float4*>(a.W1)[tid], v2 = reinterpret_cast<const float4*>(a.W2)[tid];
    float4 vi[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) vi[j] = reinterpret_cast<const float4*>(a.Wi)[j * 1024 + tid];
    {
      const int k = tid >> 4, c = (tid & 15) * 4;
      float* d1 = sW1 + k * RW_WL + c; float* d2 = sW2 + k * RW_WL + c;
      d1[0] = v1.x; d1[1] = v1.y; d1[2] = v1.z; d1[3] = v1.w;
      d2[0] = v2.x; d2[1] = v2.y; d2[2] = v2.z; d2[3] = v2.w;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int q = j * 1024 + tid, k = q / 48, c = (q - k * 48) * 4;
      float* d = sWi + k * RW_WIL + c;
      d[0] = vi[j].x; d[1] = vi[j].y; d[2] = vi[j].z; d[3] = vi[j].w;
    }
  }
  if (tid < 64) {
    sP[tid] = a.b1[tid]; sP[64 + tid] = a.g1[tid]; sP[128 + tid] = a.be1[tid];
    sP[192 + tid] = a.b2[tid]; sP[256 + tid] = a.g2[tid]; sP[320 + tid] = a.be2[tid];
  }
  if (tid < 192) sP[384 + tid] = a.bi[tid];
  __syncthreads();
  const int n_tiles = (a.M + 15) >> 4;
  const bool save = a.p1 != nullptr;
  float A[16];
  auto fetch = [&](int tile) {                   // A operand of layer 1: row li of the tile, columns lq * 16 .. + 15
    int row = tile * 16 + li;
    row = row < a.M ? row : a.M - 1;
    const float4* p = reinterpret_cast<const float4*>(a.X + (size_t)row * a.ldx + lq * 16);
#pragma unroll
    for (int j = 0; j < 4; ++j) { const float4 v = p[j]; A[4 * j] = v.x; A[4 * j + 1] = v.y; A[4 * j + 2] = v.z; A[4 * j + 3] = v.w; }
  };
  // one 64-wide layer: acc = A W + b, LayerNorm + ReLU; result to the wave's LDS tile and back as the next A operand
  auto layer = [&](const float* sW, const float* par, float* p_out, float* y_out, float* st_out, int row0) {
    f32x4 acc[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    // B fragments come from LDS: the reads of the NEXT four k-steps are in flight while the matrix cores work on the current
    // four (left to itself the compiler emits read -> wait -> 2 MFMAs with one pair of registers: the wave idles on every wait)
    float bw[2][4][4];
    auto ldw = [&](int buf, int g) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const float* wr = sW + (lq * 16 + g * 4 + kk) * RW_WL + li;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) bw[buf][kk][ct] = wr[ct * 16];
      }
    };
    ldw(0, 0);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (g + 1 < 4) ldw((g + 1) & 1, g + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[g * 4 + kk], bw[g & 1][kk][ct], acc[ct], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      const float b = par[ct * 16 + li];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc[ct][r] += b;
        s[r] += acc[ct][r];
        q[r] = fmaf(acc[ct][r], acc[ct][r], q[r]);
        const int row = row0 + lq * 4 + r;
        if (save && row < a.M) p_out[(size_t)row * FZ_H + ct * 16 + li] = acc[ct][r];
      }
    }
    float mean[4], rstd[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      mean[r] = row16_sum(s[r]) * (1.0f / 64.0f);
      const float m2 = row16_sum(q[r]) * (1.0f / 64.0f);
      rstd[r] = rsqrtf(fmaxf(m2 - mean[r] * mean[r], 0.0f) + 1e-6f);
      const int row = row0 + lq * 4 + r;
      if (save && li == 0 && row < a.M) { st_out[(size_t)row * 2] = mean[r]; st_out[(size_t)row * 2 + 1] = rstd[r]; }
    }
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      const float g = par[64 + ct * 16 + li], e = par[128 + ct * 16 + li];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float y = fmaxf((acc[ct][r] - mean[r]) * rstd[r] * g + e, 0.0f);
        sY[(lq * 4 + r) * RW_YL + ct * 16 + li] = y;
        const int row = row0 + lq * 4 + r;
        if (save && row < a.M) y_out[(size_t)row * FZ_H + ct * 16 + li] = y;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 v = *reinterpret_cast<const float4*>(sY + li * RW_YL + lq * 16 + 4 * j);
      A[4 * j] = v.x; A[4 * j + 1] = v.y; A[4 * j + 2] = v.z; A[4 * j + 3] = v.w;
    }
  };
  for (int tile = w * gridDim.x + blockIdx.x; tile < n_tiles; tile += gridDim.x * RW_WAVES) {   // consecutive tiles on different CUs
    const int row0 = tile * 16;
    fetch(tile);
    layer(sW1, sP, a.p1, a.y1, a.st1, row0);
    layer(sW2, sP + 192, a.p2, a.y2, a.st2, row0);
    // gate projection: 12 column tiles in two halves (24 accumulator registers at a time)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      f32x4 ag[6];
#pragma unroll
      for (int t = 0; t < 6; ++t) ag[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      float bg[2][2][6];                           // two k-steps per group, double-buffered (see layer())
      auto ldg = [&](int buf, int g) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const float* wr = sWi + (lq * 16 + g * 2 + kk) * RW_WIL + half * 96 + li;
#pragma unroll
          for (int t = 0; t < 6; ++t) bg[buf][kk][t] = wr[t * 16];
        }
      };
      ldg(0, 0);
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        if (g + 1 < 8) ldg((g + 1) & 1, g + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int t = 0; t < 6; ++t) ag[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[g * 2 + kk], bg[g & 1][kk][t], ag[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        const int col = half * 96 + t * 16 + li;
        const float b = sP[384 + col];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = row0 + lq * 4 + r;
          if (row < a.M) a.gi[(size_t)row * 192 + col] = ag[t][r] + b;
        }
      }
    }
  }
}

// ---- backward of the chain above: the three input gradients and the two LayerNorm+ReLU backward passes in ONE launch -----
//   dy2 = dgi Wi^T,  dpre2 = LNReLU'(p2, y2, st2, g2; dy2),  dy1 = dpre2 W2^T,  dpre1 = LNReLU'(p1, y1, st1, g1; dy1),
//   dx = dpre1 W1^T  (optionally masked by relu'(mask): the chain's input is a ReLU output for the per-agent networks)
// (jax.grad through MLP + GRUCell input Dense, dgppo/nn/mlp.py:17-29, dgppo/nn/rnn.py:14-30, as taken by
// dgppo/algo/informarl.py:377,440 and dgppo/algo/dgppo.py:316.)  It replaces five launches (dense^T, ln_relu_bwd, dense^T,
// ln_relu_bwd, dense^T) and the two round trips of dy2 / dy1 through HBM; dpre2 / dpre1 are still written because the
// weight gradients dW2 = y1^T dpre2, dW1 = x^T dpre1 (dgppo_dense_bwd_w) read them.  LayerNorm parameter gradients
// (dgamma = sum dl xhat, dbeta = sum dl) are accumulated per workgroup and added with one atomic per column.
// Same tiling as the forward: wave w owns columns 16w..16w+15 of every 64-wide result; W^T fragments (48 + 16 + 16
// registers) stay resident; 32-row tiles; the dgi tile (192 wide) is staged in LDS, the next one prefetched in registers.
struct MlpGiBwdArgs {
  const float* dgi; int M;
  const float *Wi, *W2, *W1, *g2, *g1;
  const float *p2, *y2, *st2, *p1, *y1, *st1;
  const float* mask; int ldm;
  float *dpre2, *dpre1, *dx; int lddx;
  float *dg2, *db2, *dg1, *db1;
};
#define FZ_GL 194          // LDS row stride of the dgi tile: = 2 (mod 32)
#ifndef FZ_BWD_RT
#define FZ_BWD_RT 1         // row tiles (of 16 rows) per wave in the backward chain
#endif
#ifndef FZ_BWD_WPE
#define FZ_BWD_WPE 2
#endif

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(FZ_BWD_WPE, 8))) mlp_gi_bwd_kernel(MlpGiBwdArgs a) {
  // 16-row tiles (ONE row tile per wave, the forward has two): the W^T fragments take 80 registers, every per-row quantity of
  // the two LayerNorm backward passes is live next to them, and two row tiles spill
  constexpr int BRT = FZ_BWD_RT, BRB = 16 * BRT;
  extern __shared__ float sm[];
  float* s_g = sm;                              // [RB][GL]  dgi rows
  float* s_d2 = s_g + BRB * FZ_GL;            // [RB][HL]  dpre2
  float* s_d1 = s_d2 + BRB * FZ_HL;           // [RB][HL]  dpre1
  float* s_red = s_d1 + BRB * FZ_HL;          // [RB][8]   per-row (sum dxhat, sum dxhat xhat) of each of the 4 waves
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 15, lq = lane >> 4;
  const int c = w * 16 + li;
  // B operands of X W^T: B[k][n] = W[n][k]; lane (li, lq) holds k = 4 kk + lq of output column n = c
  float wi[48], w2[16], w1[16];
#pragma unroll
  for (int kk = 0; kk < 48; ++kk) wi[kk] = a.Wi[c * 192 + kk * 4 + lq];
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) { w2[kk] = a.W2[c * FZ_H + kk * 4 + lq]; w1[kk] = a.W1[c * FZ_H + kk * 4 + lq]; }
  const float g2 = a.g2[c], g1 = a.g1[c];
  float dgs[2] = {0.f, 0.f}, dbs[2] = {0.f, 0.f};     // this lane's partial sums of dgamma / dbeta: [layer 2, layer 1], column c
  const int n_tiles = (a.M + BRB - 1) / BRB;
  constexpr int GSL = BRB * 48 / 256;       // float4 slots of a dgi tile per lane
  float4 gpf[GSL];
  auto fetch = [&](int tile) {            // BRB rows x 48 float4; rows clamped
#pragma unroll
    for (int u = 0; u < GSL; ++u) {
      const int idx = u * 256 + tid, r = idx / 48, q = idx - r * 48;
      int row = tile * BRB + r;
      row = row < a.M ? row : a.M - 1;
      gpf[u] = *reinterpret_cast<const float4*>(a.dgi + (size_t)row * 192 + 4 * q);
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int u = 0; u < GSL; ++u) {
      const int idx = u * 256 + tid, r = idx / 48, q = idx - r * 48;
      float2* d = reinterpret_cast<float2*>(s_g + r * FZ_GL + 4 * q);
      d[0] = make_float2(gpf[u].x, gpf[u].y); d[1] = make_float2(gpf[u].z, gpf[u].w);
    }
  };
  // LayerNorm + ReLU backward for this lane's 8 elements (2 row tiles x 4 rows, column c).  v: dy in, dpre out.
  auto ln_relu_bwd = [&](float (&v)[BRT][4], const float (&pv)[BRT][4], const float (&yv)[BRT][4], const float2 (&st)[BRT][4], float g,
                         float& dg_acc, float& db_acc, float* s_out, float* d_out, int row0) {
    float xh[BRT][4], dxh[BRT][4];
#pragma unroll
    for (int rt = 0; rt < BRT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rl = rt * 16 + lq * 4 + r;
        int row = row0 + rl;
        const bool ok = row < a.M;
        row = ok ? row : a.M - 1;
        const float2 ms = st[rt][r];
        xh[rt][r] = (pv[rt][r] - ms.x) * ms.y;
        const float dl = (ok && yv[rt][r] > 0.0f) ? v[rt][r] : 0.0f;
        dg_acc += dl * xh[rt][r];
        db_acc += dl;
        dxh[rt][r] = dl * g;
        const float s1 = row16_sum(dxh[rt][r]), s2 = row16_sum(dxh[rt][r] * xh[rt][r]);
        if (li == 0) { float* d = s_red + rl * 8 + 2 * w; d[0] = s1; d[1] = s2; }
        v[rt][r] = ms.y;                                  // keep rstd for the second half
      }
    FZ_LDS_BARRIER();
#pragma unroll
    for (int rt = 0; rt < BRT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rl = rt * 16 + lq * 4 + r;
        const float4 p0 = *reinterpret_cast<const float4*>(s_red + rl * 8), p1 = *reinterpret_cast<const float4*>(s_red + rl * 8 + 4);
        const float m1 = ((p0.x + p0.z) + (p1.x + p1.z)) * (1.0f / 64.0f);
        const float m2 = ((p0.y + p0.w) + (p1.y + p1.w)) * (1.0f / 64.0f);
        const float dp = v[rt][r] * (dxh[rt][r] - m1 - xh[rt][r] * m2);
        v[rt][r] = dp;
        s_out[rl * FZ_HL + c] = dp;
        const int row = row0 + rl;
        if (row < a.M) d_out[(size_t)row * FZ_H + c] = dp;
      }
    FZ_LDS_BARRIER();
  };
  auto load_st = [&](const float* src, int row0, float2 (&o)[BRT][4]) {        // (mean, rstd) of this lane's rows
#pragma unroll
    for (int rt = 0; rt < BRT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int row = row0 + rt * 16 + lq * 4 + r;
        row = row < a.M ? row : a.M - 1;
        o[rt][r] = reinterpret_cast<const float2*>(src)[row];
      }
  };
  auto load8 = [&](const float* src, int ld, int row0, float (&o)[BRT][4]) {      // this lane's 8 elements of a [M, ld] matrix
#pragma unroll
    for (int rt = 0; rt < BRT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int row = row0 + rt * 16 + lq * 4 + r;
        row = row < a.M ? row : a.M - 1;
        o[rt][r] = src[(size_t)row * ld + c];
      }
  };
  // the next tile's dgi rows are requested while the current tile computes (3 float4 per lane with 16-row tiles) and committed
  // to LDS once every read of the current tile is retired (after the first barrier pair of the first LayerNorm backward)
  int tile = blockIdx.x;
  if (tile < n_tiles) { fetch(tile); commit(); }
  __syncthreads();
  for (; tile < n_tiles; tile += gridDim.x) {
    const int row0 = tile * BRB;
    const int nxt = tile + gridDim.x;
    const bool more = nxt < n_tiles;
    if (more) fetch(nxt);
    // everything this tile reads from global memory is requested here, before the first GEMM: a load issued where it is
    // needed costs a full memory round trip per LayerNorm stage (measured: 155 -> 139 -> see profiles/README.md)
    float pv2[BRT][4], yv2[BRT][4], pv1[BRT][4], yv1[BRT][4], mk[BRT][4];
    float2 sv2[BRT][4], sv1[BRT][4];
    load8(a.p2, FZ_H, row0, pv2); load8(a.y2, FZ_H, row0, yv2); load_st(a.st2, row0, sv2);
    load8(a.p1, FZ_H, row0, pv1); load8(a.y1, FZ_H, row0, yv1); load_st(a.st1, row0, sv1);
    if (a.mask != nullptr) load8(a.mask, a.ldm, row0, mk);
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc[BRT];
    float v[BRT][4];
    // ---- dy2 = dgi Wi^T (K = 192) ----
    for (int rt = 0; rt < BRT; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c0 = 0; c0 < 48; c0 += 8) {          // chunks of 8 k-steps: 16 A registers at a time (the kernel is register-bound)
      float areg[BRT][8];
#pragma unroll
      for (int rt = 0; rt < BRT; ++rt)
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) areg[rt][kk] = s_g[(rt * 16 + li) * FZ_GL + (c0 + kk) * 4 + lq];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kk = 0; kk < 8; ++kk)
#pragma unroll
        for (int rt = 0; rt < BRT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[rt][kk], wi[c0 + kk], acc[rt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int rt = 0; rt < BRT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[rt][r] = acc[rt][r];
    // (the barriers inside ln_relu_bwd retire every read of s_g: the next tile's rows may be committed after it)
    ln_relu_bwd(v, pv2, yv2, sv2, g2, dgs[0], dbs[0], s_d2, a.dpre2, row0);
    if (more) commit();
    // ---- dy1 = dpre2 W2^T ----
    {
      float areg[BRT][16];
#pragma unroll
      for (int rt = 0; rt < BRT; ++rt)
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) areg[rt][kk] = s_d2[(rt * 16 + li) * FZ_HL + kk * 4 + lq];
      __builtin_amdgcn_sched_barrier(0);
      for (int rt = 0; rt < BRT; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 16; ++kk)
#pragma unroll
        for (int rt = 0; rt < BRT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[rt][kk], w2[kk], acc[rt], 0, 0, 0);
    }
#pragma unroll
    for (int rt = 0; rt < BRT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[rt][r] = acc[rt][r];
    ln_relu_bwd(v, pv1, yv1, sv1, g1, dgs[1], dbs[1], s_d1, a.dpre1, row0);
    // ---- dx = dpre1 W1^T ----
    {
      float areg[BRT][16];
#pragma unroll
      for (int rt = 0; rt < BRT; ++rt)
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) areg[rt][kk] = s_d1[(rt * 16 + li) * FZ_HL + kk * 4 + lq];
      __builtin_amdgcn_sched_barrier(0);
      for (int rt = 0; rt < BRT; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 16; ++kk)
#pragma unroll
        for (int rt = 0; rt < BRT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[rt][kk], w1[kk], acc[rt], 0, 0, 0);
    }
#pragma unroll
    for (int rt = 0; rt < BRT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + rt * 16 + lq * 4 + r;
        float o = acc[rt][r];
        if (a.mask != nullptr) o = (mk[rt][r] > 0.0f) ? o : 0.0f;
        if (row < a.M) a.dx[(size_t)row * a.lddx + c] = o;
      }
    // (s_d1 is next written after two more barriers of the following tile: no hazard with the reads above)
  }
  // ---- LayerNorm parameter gradients: sum the 4 row groups of each column in LDS, one atomic per column and workgroup ----
  __syncthreads();
  float* red = sm;                                // [4 values][4 lq][64 columns]
  red[(0 * 4 + lq) * 64 + c] = dgs[0]; red[(1 * 4 + lq) * 64 + c] = dbs[0];
  red[(2 * 4 + lq) * 64 + c] = dgs[1]; red[(3 * 4 + lq) * 64 + c] = dbs[1];
  __syncthreads();
  {
    const int which = tid >> 6, col = tid & 63;
    const float tot = (red[(which * 4 + 0) * 64 + col] + red[(which * 4 + 1) * 64 + col]) +
                      (red[(which * 4 + 2) * 64 + col] + red[(which * 4 + 3) * 64 + col]);
    float* dst = which == 0 ? a.dg2 : (which == 1 ? a.db2 : (which == 2 ? a.dg1 : a.db1));
    atomicAdd(dst + col, tot);
  }
}

extern "C" int32_t dgppo_mlp_gi_bwd(const float* dgi, const float* Wi, const float* W2, const float* W1, const float* g2,
                                    const float* g1, const float* p2, const float* y2, const float* st2, const float* p1,
                                    const float* y1, const float* st1, const float* relu_mask, int32_t ldm, float* dpre2,
                                    float* dpre1, float* dx, int32_t lddx, float* dg2, float* db2, float* dg1, float* db1,
                                    int32_t M, void* stream) {
  DGPPO_REQUIRE(M >= 0 && lddx >= FZ_H && (relu_mask == nullptr || ldm >= FZ_H), "mlp_gi_bwd: bad shape M=%d lddx=%d ldm=%d", M, lddx, ldm);
  DGPPO_REQUIRE(dgi && Wi && W2 && W1 && g2 && g1 && p2 && y2 && st2 && p1 && y1 && st1 && dpre2 && dpre1 && dx && dg2 && db2 &&
                dg1 && db1, "mlp_gi_bwd: NULL operand");
  DGPPO_REQUIRE((reinterpret_cast<uintptr_t>(dgi) & 15) == 0, "mlp_gi_bwd: dgi must be 16-byte aligned");
  if (M == 0) return 0;
  MlpGiBwdArgs a{dgi, M, Wi, W2, W1, g2, g1, p2, y2, st2, p1, y1, st1, relu_mask, ldm, dpre2, dpre1, dx, lddx, dg2, db2, dg1, db1};
  constexpr int BRB = 16 * FZ_BWD_RT;
  const size_t smem_t = sizeof(float) * (BRB * FZ_GL + 2 * BRB * FZ_HL + BRB * 8), smem_r = sizeof(float) * 16 * 64;
  const size_t smem = smem_t > smem_r ? smem_t : smem_r;   // the final reduction of the LayerNorm gradients reuses the buffer
  static thread_local int cap = 0;
  if (cap == 0) {
    int per_cu = 0, dev = 0, cus = 256;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(&mlp_gi_bwd_kernel), 256, smem) !=
            hipSuccess || per_cu < 1) per_cu = 1;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
    cap = per_cu * cus;
  }
  const int tiles = (M + BRB - 1) / BRB;
  hipLaunchKernelGGL(mlp_gi_bwd_kernel, dim3(tiles < cap ? tiles : cap), dim3(256), smem, (hipStream_t)stream, a);
  DGPPO_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t dgppo_mlp_gi_fwd(const float* X, int32_t ldx, const float* W1, const float* b1, const float* g1,
                                    const float* be1, const float* W2, const float* b2, const float* g2, const float* be2,
                                    const float* Wi, const float* bi, float* p1, float* y1, float* st1, float* p2, float* y2,
                                    float* st2, float* gi, int32_t M, void* stream) {
  DGPPO_REQUIRE(M >= 0 && ldx >= FZ_H, "mlp_gi_fwd: bad shape M=%d ldx=%d", M, ldx);
  DGPPO_REQUIRE(X && W1 && b1 && g1 && be1 && W2 && b2 && g2 && be2 && Wi && bi && gi, "mlp_gi_fwd: NULL operand");
  const bool save = p1 || y1 || st1 || p2 || y2 || st2;
  DGPPO_REQUIRE(!save || (p1 && y1 && st1 && p2 && y2 && st2), "mlp_gi_fwd: the saved activations are all-or-none");
  if (M == 0) return 0;
  MlpGiArgs a{X, ldx, M, W1, b1, g1, be1, W2, b2, g2, be2, Wi, bi, p1, y1, st1, p2, y2, st2, gi};
  // rows per wave, weights in LDS (one 16-wave workgroup per CU): needs 16-byte addressable input rows
  if ((ldx & 3) == 0 && ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(W1) | reinterpret_cast<uintptr_t>(W2) |
                          reinterpret_cast<uintptr_t>(Wi)) & 15) == 0 && !getenv("DGPPO_MLP_GI_TILED")) {
    static thread_local int cus = 0;
    if (cus == 0) {
      int dev = 0;
      if (hipGetDevice(&dev) != hipSuccess ||
          hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
    }
    const size_t smem_rw = sizeof(float) * (2 * 64 * RW_WL + 64 * RW_WIL + 576 + RW_WAVES * 16 * RW_YL);
    static thread_local bool attr = false;
    if (!attr) {
      const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_gi_fwd_rw_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_rw);
      DGPPO_REQUIRE(e == hipSuccess, "mlp_gi_fwd: cannot reserve %zu bytes of LDS per workgroup: %s", smem_rw, hipGetErrorString(e));
      attr = true;
    }
    const int tiles16 = (M + 15) / 16;            // spread over every CU even when there are fewer tiles than wave slots
    hipLaunchKernelGGL(mlp_gi_fwd_rw_kernel, dim3(tiles16 < cus ? tiles16 : cus), dim3(64 * RW_WAVES), smem_rw, (hipStream_t)stream, a);
    DGPPO_LAUNCH_CHECK();
    return 0;
  }
  const size_t smem = sizeof(float) * (3 * FZ_RB * FZ_HL + FZ_RB * 8);
  static thread_local int cap = 0;
  if (cap == 0) {
    int per_cu = 0, dev = 0, cus = 256;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(&mlp_gi_fwd_kernel), 256, smem) !=
            hipSuccess || per_cu < 1) per_cu = 1;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
    cap = per_cu * cus;
  }
  const int tiles = (M + FZ_RB - 1) / FZ_RB;
  hipLaunchKernelGGL(mlp_gi_fwd_kernel, dim3(tiles < cap ? tiles : cap), dim3(256), smem, (hipStream_t)stream, a);
  DGPPO_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// One GRU step (T = 1) + the output Dense layer(s) on the same rows: the tail of PPOPolicy.get_action / ValueNet.get_value
// in a rollout step (dgppo/nn/rnn.py:14-30 -> dgppo/algo/module/policy.py:62-74 [PolicyNet head Dense(64) -> TanhNormal
// Dense(4)], value.py:41,76 [Dense(n_out)]).
//   h' = GRU(gi, h0)      gi = x Wi + bi precomputed (mlp_gi_fwd_kernel),  r|z|n gate order, flax GRUCell
//   TWO = true :  u = h' W1 + b1  [64]   out = u W2 + b2   [n_out <= 16]       (policy)
//   TWO = false:  out = h' W1 + b1  [n_out <= 16]                               (value nets)
// Persistent 32-row tiles; Wh (48) + W1 (16) + W2 (16) k-step fragments stay in registers; h0 / h' / u tiles in LDS.
// ---------------------------------------------------------------------------------------------------------------------
struct GruHeadArgs {
  const float *gi, *Wh, *bhn, *h0;        // gi [M,192], Wh [64,192], bhn [64], h0 [M,64] or NULL (zeros)
  const float *W1, *b1, *W2, *b2;         // TWO: W1 [64,64], W2 [64,n_out];  else W1 [64,n_out], W2 unused
  float *hs, *hprev, *gates, *u, *out;    // hs [M,64]; hprev [M,64] / gates [M,256] / u [M,64] saved for training or NULL
  int M, n_out;
};
// GRU gate non-linearities on the hardware transcendental units: exp through v_exp_f32 (scaled by log2 e), the quotient
// through v_rcp_f32; absolute error ~1e-7 on outputs in [-1, 1], far inside the 1e-5 parity tolerance, at ~1/4 of the
// VALU instructions of expf / tanhf / IEEE division (the gate math was ~70 instructions per element).
__device__ inline float gate_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ inline float gate_tanh(float x) { return fmaf(2.0f, __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * x)), -1.0f); }


template <bool TWO>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 8))) gru1_head_fwd_kernel(GruHeadArgs a) {
  extern __shared__ float sm[];
  float* s_h = sm;                          // [RB][HL] h0 rows
  float* s_n = s_h + FZ_RB * FZ_HL;         // [RB][HL] h' rows
  float* s_u = s_n + FZ_RB * FZ_HL;         // [RB][HL] u rows (TWO)
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 15, lq = lane >> 4;
  const int c = w * 16 + li;
  float wh[16][3], w1[16], w2[16];
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) {
    const int k = kk * 4 + lq;
#pragma unroll
    for (int t = 0; t < 3; ++t) wh[kk][t] = a.Wh[k * 192 + t * 64 + c];
    if (TWO) {
      w1[kk] = a.W1[k * FZ_H + c];
      w2[kk] = (li < a.n_out) ? a.W2[k * a.n_out + li] : 0.0f;      // single column tile: identical in every wave
    } else {
      w1[kk] = (li < a.n_out) ? a.W1[k * a.n_out + li] : 0.0f;
      w2[kk] = 0.0f;
    }
  }
  const float bn = a.bhn[c];
  const float b1 = TWO ? a.b1[c] : ((li < a.n_out) ? a.b1[li] : 0.0f);
  const float b2 = (TWO && li < a.n_out) ? a.b2[li] : 0.0f;
  const int n_tiles = (a.M + FZ_RB - 1) / FZ_RB;
  float hpf[2][4];
  auto fetch_h0 = [&](int tile) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int idx = u * 256 + tid, r = idx >> 4, q = idx & 15;
      int row = tile * FZ_RB + r;
      row = row < a.M ? row : a.M - 1;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (a.h0 != nullptr) v = reinterpret_cast<const float4*>(a.h0 + (size_t)row * FZ_H)[q];
      hpf[u][0] = v.x; hpf[u][1] = v.y; hpf[u][2] = v.z; hpf[u][3] = v.w;
    }
  };
  auto commit_h0 = [&]() {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int idx = u * 256 + tid, r = idx >> 4, q = idx & 15;
      float* d = s_h + r * FZ_HL + 4 * q;
      d[0] = hpf[u][0]; d[1] = hpf[u][1]; d[2] = hpf[u][2]; d[3] = hpf[u][3];
    }
  };
  int tile = blockIdx.x;
  if (tile < n_tiles) { fetch_h0(tile); commit_h0(); }
  __syncthreads();
  for (; tile < n_tiles; tile += gridDim.x) {
    const int row0 = tile * FZ_RB;
    const int nxt = tile + gridDim.x;
    const bool more = nxt < n_tiles;
    // gate inputs of this tile and the next tile's h0 rows: requested before the MFMAs
    float g[2][4][3];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int row = row0 + rt * 16 + lq * 4 + r;
        row = row < a.M ? row : a.M - 1;
        const float* gp = a.gi + (size_t)row * 192 + c;
        g[rt][r][0] = gp[0]; g[rt][r][1] = gp[64]; g[rt][r][2] = gp[128];
      }
    if (more) fetch_h0(nxt);
    __builtin_amdgcn_sched_barrier(0);
    float areg[2][16];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) areg[rt][kk] = s_h[(rt * 16 + li) * FZ_HL + kk * 4 + lq];
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc[2][3];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int t = 0; t < 3; ++t) acc[rt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 16; ++kk)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int t = 0; t < 3; ++t) acc[rt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[rt][kk], wh[kk][t], acc[rt][t], 0, 0, 0);
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rl = rt * 16 + lq * 4 + r, row = row0 + rl;
        const float hp = s_h[rl * FZ_HL + c];
        const float rg = gate_sigmoid(g[rt][r][0] + acc[rt][0][r]);
        const float zg = gate_sigmoid(g[rt][r][1] + acc[rt][1][r]);
        const float hn = acc[rt][2][r] + bn;
        const float ng = gate_tanh(g[rt][r][2] + rg * hn);
        const float hnew = (1.0f - zg) * ng + zg * hp;
        s_n[rl * FZ_HL + c] = hnew;
        if (row < a.M) {
          a.hs[(size_t)row * FZ_H + c] = hnew;
          if (a.hprev != nullptr) a.hprev[(size_t)row * FZ_H + c] = hp;
          if (a.gates != nullptr) {
            float* gs = a.gates + (size_t)row * 256;
            gs[c] = rg; gs[64 + c] = zg; gs[128 + c] = ng; gs[192 + c] = hn;
          }
        }
      }
    FZ_LDS_BARRIER();                     // h' complete; every read of s_h is done
    if (more) commit_h0();
    // ---- first Dense on h' ----
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) areg[rt][kk] = s_n[(rt * 16 + li) * FZ_HL + kk * 4 + lq];
    __builtin_amdgcn_sched_barrier(0);
    if (TWO) {
      f32x4 au[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int kk = 0; kk < 16; ++kk)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) au[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[rt][kk], w1[kk], au[rt], 0, 0, 0);
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int rl = rt * 16 + lq * 4 + r, row = row0 + rl;
          const float uv = au[rt][r] + b1;
          s_u[rl * FZ_HL + c] = uv;
          if (a.u != nullptr && row < a.M) a.u[(size_t)row * FZ_H + c] = uv;
        }
      FZ_LDS_BARRIER();
      // ---- second Dense: one column tile, row tile w (waves 0 and 1) ----
      if (w < 2) {
        float ar[16];
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) ar[kk] = s_u[(w * 16 + li) * FZ_HL + kk * 4 + lq];
        f32x4 ao = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) ao = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[kk], w2[kk], ao, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = row0 + w * 16 + lq * 4 + r;
          if (row < a.M && li < a.n_out) a.out[(size_t)row * a.n_out + li] = ao[r] + b2;
        }
      }
    } else if (w < 2) {
      f32x4 ao = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) ao = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[w][kk], w1[kk], ao, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + w * 16 + lq * 4 + r;
        if (row < a.M && li < a.n_out) a.out[(size_t)row * a.n_out + li] = ao[r] + b1;
      }
    }
    FZ_LDS_BARRIER();                     // next tile's h0 committed; s_n / s_u free again
  }
}

// ---- the same step with rows per WAVE and the weights in LDS (inference form: nothing saved for a backward) -----------------
// As mlp_gi_fwd_rw_kernel: a wave carries 16 rows through the GRU step and the head; Wh (48 KB), W1 and W2 sit in LDS once per CU,
// shared by the 16 waves of the one workgroup; h' and u change from the C/D layout to the A layout through a wave-private LDS
// tile; no barrier after the staging.  The rollout step (32 768 rows) took 19 us with the tiled kernel above (two workgroup
// barriers per 32-row tile, 212 VGPRs: two workgroups per CU) for 9 us of matrix-core work.
#define RWH_W1L 65
template <bool TWO>
__global__ void __launch_bounds__(64 * RW_WAVES) gru1_head_fwd_rw_kernel(GruHeadArgs a) {
  extern __shared__ float sm[];
  float* sWh = sm;                                // [64][193]
  float* sW1 = sWh + 64 * RW_WIL;                 // TWO: [64][65] (64 columns); else [64][17] in the same stride-65 rows (16 columns)
  float* sW2 = sW1 + 64 * RWH_W1L;                // TWO: [64][17]
  float* sB = sW2 + 64 * 17;                      // bhn (64), b1 (64), b2 (16)
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 15, lq = lane >> 4;
  float* sY = sB + 144 + w * (16 * RW_YL);
  {
    static_assert(RW_WAVES == 16, "the staging below assumes 1024 threads");
    float4 vh[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) vh[j] = reinterpret_cast<const float4*>(a.Wh)[j * 1024 + tid];
    float w1v[4] = {0.f, 0.f, 0.f, 0.f};
    float w2v = 0.0f;
    if (TWO) {
      const float4 v = reinterpret_cast<const float4*>(a.W1)[tid];          // [64][64]
      w1v[0] = v.x; w1v[1] = v.y; w1v[2] = v.z; w1v[3] = v.w;
      { const int k = tid >> 4, c = tid & 15; w2v = (c < a.n_out) ? a.W2[k * a.n_out + c] : 0.0f; }
    } else {
      const int k = tid >> 4, c = tid & 15;
      w1v[0] = (c < a.n_out) ? a.W1[k * a.n_out + c] : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int q = j * 1024 + tid, k = q / 48, c = (q - k * 48) * 4;
      float* d = sWh + k * RW_WIL + c;
      d[0] = vh[j].x; d[1] = vh[j].y; d[2] = vh[j].z; d[3] = vh[j].w;
    }
    if (TWO) {
      const int k = tid >> 4, c = (tid & 15) * 4;
      float* d = sW1 + k * RWH_W1L + c;
      d[0] = w1v[0]; d[1] = w1v[1]; d[2] = w1v[2]; d[3] = w1v[3];
      sW2[(tid >> 4) * 17 + (tid & 15)] = w2v;
    } else {
      sW1[(tid >> 4) * RWH_W1L + (tid & 15)] = w1v[0];
    }
    if (tid < 64) { sB[tid] = a.bhn[tid]; sB[64 + tid] = TWO ? a.b1[tid] : ((tid < a.n_out) ? a.b1[tid] : 0.0f); }
    if (tid < 16) sB[128 + tid] = (TWO && tid < a.n_out) ? a.b2[tid] : 0.0f;
  }
  __syncthreads();
  const int n_tiles = (a.M + 15) >> 4;
  for (int tile = w * gridDim.x + blockIdx.x; tile < n_tiles; tile += gridDim.x * RW_WAVES) {
    const int row0 = tile * 16;
    // A operand: h0 row li, columns lq * 16 .. + 15 (zeros without h0); the same rows again in the C/D layout for the update
    float A[16];
    {
      int row = row0 + li;
      row = row < a.M ? row : a.M - 1;
      if (a.h0 != nullptr) {
        const float4* p = reinterpret_cast<const float4*>(a.h0 + (size_t)row * FZ_H + lq * 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float4 v = p[j]; A[4 * j] = v.x; A[4 * j + 1] = v.y; A[4 * j + 2] = v.z; A[4 * j + 3] = v.w; }
      } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) A[j] = 0.0f;
      }
    }
    float hp[4][4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int row = row0 + lq * 4 + r;
        row = row < a.M ? row : a.M - 1;
        hp[ct][r] = (a.h0 != nullptr) ? a.h0[(size_t)row * FZ_H + ct * 16 + li] : 0.0f;
      }
    // hidden column group ct (16 columns): r, z, n gate tiles ct, ct + 4, ct + 8 of gh = h0 Wh, seeded with the input projection gi
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      f32x4 ar, az, an = f32x4{0.f, 0.f, 0.f, 0.f};
      float gn[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int row = row0 + lq * 4 + r;
        row = row < a.M ? row : a.M - 1;
        const float* gp = a.gi + (size_t)row * 192 + ct * 16 + li;
        ar[r] = gp[0]; az[r] = gp[64]; gn[r] = gp[128];
      }
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) {
        const float* wr = sWh + (lq * 16 + kk) * RW_WIL + ct * 16 + li;
        ar = __builtin_amdgcn_mfma_f32_16x16x4f32(A[kk], wr[0], ar, 0, 0, 0);
        az = __builtin_amdgcn_mfma_f32_16x16x4f32(A[kk], wr[64], az, 0, 0, 0);
        an = __builtin_amdgcn_mfma_f32_16x16x4f32(A[kk], wr[128], an, 0, 0, 0);
      }
      const float bn = sB[ct * 16 + li];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float rg = gate_sigmoid(ar[r]);
        const float zg = gate_sigmoid(az[r]);
        const float ng = gate_tanh(gn[r] + rg * (an[r] + bn));
        const float hnew = (1.0f - zg) * ng + zg * hp[ct][r];
        sY[(lq * 4 + r) * RW_YL + ct * 16 + li] = hnew;
        const int row = row0 + lq * 4 + r;
        if (row < a.M) a.hs[(size_t)row * FZ_H + ct * 16 + li] = hnew;
      }
    }
    // h' as the next A operand
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 v = *reinterpret_cast<const float4*>(sY + li * RW_YL + lq * 16 + 4 * j);
      A[4 * j] = v.x; A[4 * j + 1] = v.y; A[4 * j + 2] = v.z; A[4 * j + 3] = v.w;
    }
    if (TWO) {
      f32x4 au[4];
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) au[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) {
        const float* wr = sW1 + (lq * 16 + kk) * RWH_W1L + li;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) au[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[kk], wr[ct * 16], au[ct], 0, 0, 0);
      }
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const float b = sB[64 + ct * 16 + li];
#pragma unroll
        for (int r = 0; r < 4; ++r) sY[(lq * 4 + r) * RW_YL + ct * 16 + li] = au[ct][r] + b;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 v = *reinterpret_cast<const float4*>(sY + li * RW_YL + lq * 16 + 4 * j);
        A[4 * j] = v.x; A[4 * j + 1] = v.y; A[4 * j + 2] = v.z; A[4 * j + 3] = v.w;
      }
    }
    {
      f32x4 ao = f32x4{0.f, 0.f, 0.f, 0.f};
      const float* sWo = TWO ? sW2 : sW1;
      const int ld = TWO ? 17 : RWH_W1L;
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) ao = __builtin_amdgcn_mfma_f32_16x16x4f32(A[kk], sWo[(lq * 16 + kk) * ld + li], ao, 0, 0, 0);
      const float b = TWO ? sB[128 + li] : sB[64 + li];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + lq * 4 + r;
        if (row < a.M && li < a.n_out) a.out[(size_t)row * a.n_out + li] = ao[r] + b;
      }
    }
  }
}

extern "C" int32_t dgppo_gru1_head_fwd(const float* gi, const float* Wh, const float* bhn, const float* h0, const float* W1,
                                       const float* b1, const float* W2, const float* b2, float* hs, float* hprev,
                                       float* gates, float* u, float* out, int32_t M, int32_t n_out, void* stream) {
  DGPPO_REQUIRE(M >= 0 && n_out >= 1 && n_out <= 16, "gru1_head_fwd: bad shape M=%d n_out=%d", M, n_out);
  DGPPO_REQUIRE(gi && Wh && bhn && W1 && b1 && hs && out, "gru1_head_fwd: NULL operand");
  const bool two = W2 != nullptr;
  DGPPO_REQUIRE(!two || b2, "gru1_head_fwd: W2 without b2");
  DGPPO_REQUIRE(two || !u, "gru1_head_fwd: u is only produced by the two-layer head");
  if (M == 0) return 0;
  GruHeadArgs a{gi, Wh, bhn, h0, W1, b1, W2, b2, hs, hprev, gates, u, out, M, n_out};
  // inference (nothing saved): rows per wave, weights in LDS
  if (!hprev && !gates && !u && ((reinterpret_cast<uintptr_t>(Wh) | reinterpret_cast<uintptr_t>(W1) | reinterpret_cast<uintptr_t>(h0)) & 15) == 0 &&
      !getenv("DGPPO_GRU1_TILED")) {
    static thread_local int cus = 0;
    if (cus == 0) {
      int dev = 0;
      if (hipGetDevice(&dev) != hipSuccess ||
          hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
    }
    const size_t smem_rw = sizeof(float) * (64 * RW_WIL + 64 * RWH_W1L + 64 * 17 + 144 + RW_WAVES * 16 * RW_YL);
    static thread_local bool attr[2] = {false, false};
    if (!attr[two]) {
      const void* f = two ? reinterpret_cast<const void*>(&gru1_head_fwd_rw_kernel<true>)
                          : reinterpret_cast<const void*>(&gru1_head_fwd_rw_kernel<false>);
      const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_rw);
      DGPPO_REQUIRE(e == hipSuccess, "gru1_head_fwd: cannot reserve %zu bytes of LDS per workgroup: %s", smem_rw, hipGetErrorString(e));
      attr[two] = true;
    }
    const int tiles16 = (M + 15) / 16, grid_rw = tiles16 < cus ? tiles16 : cus;
    if (two) hipLaunchKernelGGL(gru1_head_fwd_rw_kernel<true>, dim3(grid_rw), dim3(64 * RW_WAVES), smem_rw, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(gru1_head_fwd_rw_kernel<false>, dim3(grid_rw), dim3(64 * RW_WAVES), smem_rw, (hipStream_t)stream, a);
    DGPPO_LAUNCH_CHECK();
    return 0;
  }
  const size_t smem = sizeof(float) * 3 * FZ_RB * FZ_HL;
  const void* fn = two ? reinterpret_cast<const void*>(&gru1_head_fwd_kernel<true>)
                       : reinterpret_cast<const void*>(&gru1_head_fwd_kernel<false>);
  static thread_local int cap[2] = {0, 0};
  if (cap[two] == 0) {
    int per_cu = 0, dev = 0, cus = 256;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, smem) != hipSuccess || per_cu < 1) per_cu = 1;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
    cap[two] = per_cu * cus;
  }
  const int tiles = (M + FZ_RB - 1) / FZ_RB;
  const int grid = tiles < cap[two] ? tiles : cap[two];
  if (two) hipLaunchKernelGGL(gru1_head_fwd_kernel<true>, dim3(grid), dim3(256), smem, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(gru1_head_fwd_kernel<false>, dim3(grid), dim3(256), smem, (hipStream_t)stream, a);
  DGPPO_LAUNCH_CHECK();
  return 0;
}
