// Weight preparation for the fixed-fan-in GNN layer (see nn_graph.hip) and its adjoint.
// From the reference's GraphTransformer parameters (dgppo/nn/gnn.py:86-110; flax names Dense_0..4 = q,k,v,e,u):
//   Mcat[f, h*Fp+g] = 1/sqrt(D) * sum_d Wq[f,hD+d] Wk[g,hD+d]      cvec[h*Fp+g] = 1/sqrt(D) * sum_d bq[hD+d] Wk[g,hD+d]
//   Wout = [ Wu ; per head (Wv_h ; We_h)/H ; mean_h bv_h ; 0 ]       (rows padded to Kp)
// so that  logits = (x_i Mcat + cvec) . x_s   and   x_i' = relu([x_i | agg | 1] Wout + bu).
// The key bias bk only shifts all logits of one (receiver, head) and cancels in the softmax: its gradient is exactly 0.
#include "common.h"

struct PrepArgs {
  const float *Wq, *bq, *Wk, *Wv, *bv, *We, *Wu;
  float *Mcat, *cvec, *Wout;
  // adjoint
  const float *dMcat, *dcvec, *dWout;
  float *dWq, *dbq, *dWk, *dWv, *dbv, *dWe, *dWu;
  int F, Fp, D, H, Kp;
};

__global__ void gnn_prep_kernel(PrepArgs a) {
  const int F = a.F, Fp = a.Fp, D = a.D, H = a.H, HD = H * D, W = Fp + 4;
  const float sc = 1.0f / sqrtf((float)D);
  const int nM = Fp * H * Fp, nC = H * Fp, nW = a.Kp * D;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < nM + nC + nW; idx += gridDim.x * blockDim.x) {
    if (idx < nM) {
      const int f = idx / (H * Fp), hg = idx - f * (H * Fp), h = hg / Fp, g = hg - h * Fp;
      float acc = 0.0f;
      if (f < F && g < F)
        for (int d = 0; d < D; ++d) acc = fmaf(a.Wq[f * HD + h * D + d], a.Wk[g * HD + h * D + d], acc);
      a.Mcat[idx] = acc * sc;
    } else if (idx < nM + nC) {
      const int hg = idx - nM, h = hg / Fp, g = hg - h * Fp;
      float acc = 0.0f;
      if (g < F)
        for (int d = 0; d < D; ++d) acc = fmaf(a.bq[h * D + d], a.Wk[g * HD + h * D + d], acc);
      a.cvec[hg] = acc * sc;
    } else {
      const int o = idx - nM - nC, row = o / D, d = o - row * D;
      const int kc = Fp + H * W;
      float v = 0.0f;
      if (row < Fp) v = (row < F) ? a.Wu[row * D + d] : 0.0f;
      else if (row < kc) {
        const int q = row - Fp, h = q / W, w = q - h * W;
        if (w < Fp) v = (w < F) ? a.Wv[w * HD + h * D + d] / (float)H : 0.0f;
        else v = a.We[(w - Fp) * HD + h * D + d] / (float)H;
      } else if (row == kc) {
        float acc = 0.0f;
        for (int h = 0; h < H; ++h) acc += a.bv[h * D + d];
        v = acc / (float)H;
      }
      a.Wout[o] = v;
    }
  }
}

__global__ void gnn_unprep_kernel(PrepArgs a) {
  const int F = a.F, Fp = a.Fp, D = a.D, H = a.H, HD = H * D, W = Fp + 4;
  const float sc = 1.0f / sqrtf((float)D);
  const int kc = Fp + H * W;
  const int nQ = F * HD, nB = HD, nE = 4 * HD, nU = F * D;
  // ranges: dWq | dbq | dWk | dWv | dbv | dWe | dWu
  const int total = nQ + nB + nQ + nQ + nB + nE + nU;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    int o = idx;
    if (o < nQ) {  // dWq[f, hD+d] = sc * sum_g dMcat[f, hFp+g] Wk[g, hD+d]
      const int f = o / HD, hd = o - f * HD, h = hd / D;
      float acc = 0.0f;
      for (int g = 0; g < F; ++g) acc = fmaf(a.dMcat[f * (H * Fp) + h * Fp + g], a.Wk[g * HD + hd], acc);
      a.dWq[o] += acc * sc;
      continue;
    }
    o -= nQ;
    if (o < nB) {  // dbq[hD+d] = sc * sum_g dcvec[hFp+g] Wk[g, hD+d]
      const int h = o / D;
      float acc = 0.0f;
      for (int g = 0; g < F; ++g) acc = fmaf(a.dcvec[h * Fp + g], a.Wk[g * HD + o], acc);
      a.dbq[o] += acc * sc;
      continue;
    }
    o -= nB;
    if (o < nQ) {  // dWk[g, hD+d] = sc * (sum_f dMcat[f, hFp+g] Wq[f, hD+d] + dcvec[hFp+g] bq[hD+d])
      const int g = o / HD, hd = o - g * HD, h = hd / D;
      float acc = a.dcvec[h * Fp + g] * a.bq[hd];
      for (int f = 0; f < F; ++f) acc = fmaf(a.dMcat[f * (H * Fp) + h * Fp + g], a.Wq[f * HD + hd], acc);
      a.dWk[o] += acc * sc;
      continue;
    }
    o -= nQ;
    if (o < nQ) {  // dWv
      const int f = o / HD, hd = o - f * HD, h = hd / D, d = hd - h * D;
      a.dWv[o] += a.dWout[(Fp + h * W + f) * D + d] / (float)H;
      continue;
    }
    o -= nQ;
    if (o < nB) {  // dbv
      const int d = o % D;
      a.dbv[o] += a.dWout[kc * D + d] / (float)H;
      continue;
    }
    o -= nB;
    if (o < nE) {  // dWe
      const int c = o / HD, hd = o - c * HD, h = hd / D, d = hd - h * D;
      a.dWe[o] += a.dWout[(Fp + h * W + Fp + c) * D + d] / (float)H;
      continue;
    }
    o -= nE;
    a.dWu[o] += a.dWout[o];  // rows [0,F) x D, same flat index since both are [*, D]
  }
}

static int32_t prep_check(const PrepArgs& a) {
  DGPPO_REQUIRE(a.F >= 1 && a.Fp >= a.F && a.Fp <= 64 && a.D >= 1 && a.D <= 128 && a.H >= 1 && a.H <= 8, "gnn_prep: bad dims");
  DGPPO_REQUIRE(a.Kp >= a.Fp + a.H * (a.Fp + 4) + 1, "gnn_prep: Kp too small");
  return 0;
}

extern "C" int32_t dgppo_gnn_prep(const float* Wq, const float* bq, const float* Wk, const float* Wv, const float* bv,
                                  const float* We, const float* Wu, float* Mcat, float* cvec, float* Wout, int32_t F,
                                  int32_t Fp, int32_t D, int32_t H, int32_t Kp, void* stream) {
  PrepArgs a{};
  a.Wq = Wq; a.bq = bq; a.Wk = Wk; a.Wv = Wv; a.bv = bv; a.We = We; a.Wu = Wu; a.Mcat = Mcat; a.cvec = cvec; a.Wout = Wout;
  a.F = F; a.Fp = Fp; a.D = D; a.H = H; a.Kp = Kp;
  int32_t rc = prep_check(a);
  if (rc) return rc;
  DGPPO_REQUIRE(Wq && bq && Wk && Wv && bv && We && Wu && Mcat && cvec && Wout, "gnn_prep: NULL operand");
  hipLaunchKernelGGL(gnn_prep_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, a);
  DGPPO_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t dgppo_gnn_unprep(const float* dMcat, const float* dcvec, const float* dWout, const float* Wq,
                                    const float* bq, const float* Wk, float* dWq, float* dbq, float* dWk, float* dWv,
                                    float* dbv, float* dWe, float* dWu, int32_t F, int32_t Fp, int32_t D, int32_t H,
                                    int32_t Kp, void* stream) {
  PrepArgs a{};
  a.dMcat = dMcat; a.dcvec = dcvec; a.dWout = dWout; a.Wq = Wq; a.bq = bq; a.Wk = Wk;
  a.dWq = dWq; a.dbq = dbq; a.dWk = dWk; a.dWv = dWv; a.dbv = dbv; a.dWe = dWe; a.dWu = dWu;
  a.F = F; a.Fp = Fp; a.D = D; a.H = H; a.Kp = Kp;
  int32_t rc = prep_check(a);
  if (rc) return rc;
  DGPPO_REQUIRE(dMcat && dcvec && dWout && Wq && bq && Wk && dWq && dbq && dWk && dWv && dbv && dWe && dWu,
                "gnn_unprep: NULL operand");
  hipLaunchKernelGGL(gnn_unprep_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, a);
  DGPPO_LAUNCH_CHECK();
  return 0;
}
