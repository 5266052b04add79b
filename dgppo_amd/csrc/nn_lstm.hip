// LSTM scan / BPTT for the --use-lstm option (dgppo/nn/rnn.py:22-24 with flax nn.LSTMCell(features=64)):
//   i = s(W_ii x + W_hi h + b_hi)   f = s(W_if x + W_hf h + b_hf)   g = tanh(W_ig x + W_hg h + b_hg)   o = s(W_io x + W_ho h + b_ho)
//   c' = f c + i g        h' = o tanh(c')        carry = (c, h), output = h'          [upstream flax, SURVEY A.6 style]
// The input projection zi = x [W_ii | W_if | W_ig | W_io] (no bias in flax's input Denses) is a plain Dense outside; this
// file does the recurrent part over T steps and its backward.  An option off the benchmark path: a straightforward VALU
// kernel — one workgroup per 4 sequences, thread = (sequence, hidden unit), W_h (64 x 256 fp32 = 64 KiB) staged in LDS once
// per workgroup, h of the 4 sequences exchanged through LDS every step.  Row addressing as dgppo_gru_fwd: sequence s at
// step tau lives in row ((s / n_inner) * T + tau) * n_inner + s % n_inner.
#include "common.h"

#define LH 64
#define LG 256
#define LSEQ 4

struct LstmArgs {
  const float* zi;      // [rows, 256] input projection (i | f | g | o)
  const float* Wh;      // [64, 256]
  const float* bh;      // [256]
  const float* c0;      // [n_seq, 64] or NULL
  const float* h0;      // [n_seq, 64] or NULL
  float* cs;            // [rows, 64] cell state after each step
  float* hs;            // [rows, 64] output after each step
  float* cprev;         // [rows, 64] saved for the backward (may be NULL)
  float* hprev;         // [rows, 64]
  float* gates;         // [rows, 256] post-activation i | f | g | o
  // backward
  const float* dhs;     // [rows, 64] gradient of every step's output
  float* dz;            // [rows, 256] gradient of the pre-activations
  int n_seq, T, n_inner;
};

__device__ inline float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ inline long lstm_row(int s, int tau, int T, int n_inner) {
  return ((long)(s / n_inner) * T + tau) * n_inner + (s % n_inner);
}

__global__ void __launch_bounds__(256) lstm_fwd_kernel(LstmArgs a) {
  extern __shared__ float sm[];
  float* s_W = sm;                    // [64][256]
  float* s_h = s_W + LH * LG;         // [LSEQ][64]
  const int tid = threadIdx.x, sl = tid >> 6, j = tid & 63;
  for (int i = tid; i < LH * LG; i += 256) s_W[i] = a.Wh[i];
  const float bi = a.bh[j], bf = a.bh[64 + j], bg = a.bh[128 + j], bo = a.bh[192 + j];
  for (int s0 = blockIdx.x * LSEQ; s0 < a.n_seq; s0 += gridDim.x * LSEQ) {
    const int s = s0 + sl;
    const bool live = s < a.n_seq;
    float c = (live && a.c0) ? a.c0[(size_t)s * LH + j] : 0.0f;
    float h = (live && a.h0) ? a.h0[(size_t)s * LH + j] : 0.0f;
    for (int tau = 0; tau < a.T; ++tau) {
      __syncthreads();
      s_h[sl * LH + j] = h;
      __syncthreads();
      if (!live) continue;
      const long r = lstm_row(s, tau, a.T, a.n_inner);
      float zi = a.zi[r * LG + j] + bi, zf = a.zi[r * LG + 64 + j] + bf, zg = a.zi[r * LG + 128 + j] + bg,
            zo = a.zi[r * LG + 192 + j] + bo;
      const float* hv = s_h + sl * LH;
#pragma unroll 8
      for (int k = 0; k < LH; ++k) {
        const float hk = hv[k];
        const float* w = s_W + k * LG;
        zi = fmaf(hk, w[j], zi); zf = fmaf(hk, w[64 + j], zf); zg = fmaf(hk, w[128 + j], zg); zo = fmaf(hk, w[192 + j], zo);
      }
      const float gi = sigm(zi), gf = sigm(zf), gg = tanhf(zg), go = sigm(zo);
      if (a.cprev) { a.cprev[r * LH + j] = c; a.hprev[r * LH + j] = h; }
      if (a.gates) { a.gates[r * LG + j] = gi; a.gates[r * LG + 64 + j] = gf; a.gates[r * LG + 128 + j] = gg; a.gates[r * LG + 192 + j] = go; }
      c = gf * c + gi * gg;
      h = go * tanhf(c);
      a.cs[r * LH + j] = c;
      a.hs[r * LH + j] = h;
    }
  }
}

// BPTT: dz of every step from dhs (gradient of the outputs; the final carry has no gradient: sequences end at the chunk)
__global__ void __launch_bounds__(256) lstm_bwd_kernel(LstmArgs a) {
  extern __shared__ float sm[];
  float* s_W = sm;                    // [64][256]
  float* s_dz = s_W + LH * LG;        // [LSEQ][256]
  const int tid = threadIdx.x, sl = tid >> 6, j = tid & 63;
  for (int i = tid; i < LH * LG; i += 256) s_W[i] = a.Wh[i];
  for (int s0 = blockIdx.x * LSEQ; s0 < a.n_seq; s0 += gridDim.x * LSEQ) {
    const int s = s0 + sl;
    const bool live = s < a.n_seq;
    float dh_rec = 0.0f, dc = 0.0f;
    for (int tau = a.T - 1; tau >= 0; --tau) {
      float dzi = 0.f, dzf = 0.f, dzg = 0.f, dzo = 0.f;
      long r = 0;
      if (live) {
        r = lstm_row(s, tau, a.T, a.n_inner);
        const float gi = a.gates[r * LG + j], gf = a.gates[r * LG + 64 + j], gg = a.gates[r * LG + 128 + j],
                    go = a.gates[r * LG + 192 + j];
        const float cp = a.cprev[r * LH + j];
        const float cn = gf * cp + gi * gg, tc = tanhf(cn);
        const float dh = a.dhs[r * LH + j] + dh_rec;
        const float dcn = dc + dh * go * (1.0f - tc * tc);
        dzo = dh * tc * go * (1.0f - go);
        dzi = dcn * gg * gi * (1.0f - gi);
        dzf = dcn * cp * gf * (1.0f - gf);
        dzg = dcn * gi * (1.0f - gg * gg);
        dc = dcn * gf;
        a.dz[r * LG + j] = dzi; a.dz[r * LG + 64 + j] = dzf; a.dz[r * LG + 128 + j] = dzg; a.dz[r * LG + 192 + j] = dzo;
      }
      __syncthreads();
      s_dz[sl * LG + j] = dzi; s_dz[sl * LG + 64 + j] = dzf; s_dz[sl * LG + 128 + j] = dzg; s_dz[sl * LG + 192 + j] = dzo;
      __syncthreads();
      // dh_{tau-1}[j] = sum_q dz[q] Wh[j, q]
      float acc = 0.0f;
      const float* w = s_W + j * LG;
      const float* d = s_dz + sl * LG;
#pragma unroll 8
      for (int q = 0; q < LG; ++q) acc = fmaf(d[q], w[q], acc);
      dh_rec = acc;
    }
  }
}

static int lstm_launch(const LstmArgs& a, bool bwd, hipStream_t s) {
  const size_t smem = sizeof(float) * (LH * LG + (bwd ? LSEQ * LG : LSEQ * LH));
  static bool attr_done[2] = {false, false};
  if (!attr_done[bwd]) {
    (void)hipFuncSetAttribute(bwd ? (const void*)lstm_bwd_kernel : (const void*)lstm_fwd_kernel,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_done[bwd] = true;
  }
  int tiles = (a.n_seq + LSEQ - 1) / LSEQ;
  const int grid = tiles < 512 ? tiles : 512;
  if (bwd) hipLaunchKernelGGL(lstm_bwd_kernel, dim3(grid), dim3(256), smem, s, a);
  else hipLaunchKernelGGL(lstm_fwd_kernel, dim3(grid), dim3(256), smem, s, a);
  return 0;
}

extern "C" int32_t dgppo_lstm_fwd(const float* zi, const float* Wh, const float* bh, const float* c0, const float* h0,
                                  float* cs, float* hs, float* cprev, float* hprev, float* gates, int32_t n_seq,
                                  int32_t T, int32_t n_inner, void* stream) {
  DGPPO_REQUIRE(n_seq >= 0 && T >= 1 && n_inner >= 1, "lstm_fwd: bad sizes");
  if (n_seq == 0) return 0;
  DGPPO_REQUIRE(n_seq % n_inner == 0, "lstm_fwd: n_seq must be a multiple of n_inner");
  DGPPO_REQUIRE(zi && Wh && bh && cs && hs, "lstm_fwd: NULL operand");
  DGPPO_REQUIRE((cprev == nullptr) == (hprev == nullptr), "lstm_fwd: cprev and hprev come together");
  LstmArgs a{zi, Wh, bh, c0, h0, cs, hs, cprev, hprev, gates, nullptr, nullptr, n_seq, T, n_inner};
  lstm_launch(a, false, (hipStream_t)stream);
  DGPPO_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t dgppo_lstm_bwd(const float* dhs, const float* Wh, const float* cprev, const float* gates, float* dz,
                                  int32_t n_seq, int32_t T, int32_t n_inner, void* stream) {
  DGPPO_REQUIRE(n_seq >= 0 && T >= 1 && n_inner >= 1, "lstm_bwd: bad sizes");
  if (n_seq == 0) return 0;
  DGPPO_REQUIRE(n_seq % n_inner == 0, "lstm_bwd: n_seq must be a multiple of n_inner");
  DGPPO_REQUIRE(dhs && Wh && cprev && gates && dz, "lstm_bwd: NULL operand");
  LstmArgs a{nullptr, Wh, nullptr, nullptr, nullptr, nullptr, nullptr, const_cast<float*>(cprev), nullptr,
             const_cast<float*>(gates), dhs, dz, n_seq, T, n_inner};
  lstm_launch(a, true, (hipStream_t)stream);
  DGPPO_LAUNCH_CHECK();
  return 0;
}
