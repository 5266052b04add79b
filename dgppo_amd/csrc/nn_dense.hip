// Dense (fully connected) building blocks on the fp32 matrix cores of gfx950 (v_mfma_f32_16x16x4_f32):
//   dgppo_dense_fwd   Y[M,N]  = act(X[M,K] * W[K,N] + b)        (also  dX = dY * W^T via trans_w)
//   dgppo_dense_bwd_w dW[K,N] += X[M,K]^T * dY[M,N],  db[N] += colsum(dY)
// They replace flax nn.Dense as used by dgppo/nn/mlp.py:19-22, dgppo/nn/gnn.py:86-110, dgppo/nn/rnn.py:19-21 (GRUCell
// input/recurrent projections), dgppo/algo/module/policy.py:67-70 and dgppo/algo/module/value.py:41,76 — and the
// jax.grad of those (dgppo/algo/informarl.py:377,440; dgppo/algo/dgppo.py:316).
// fp32-input MFMA is bit-for-bit a k-ordered fmaf chain (exact fp32, no reduced precision).
#include "common.h"

using f32x4 = __attribute__((ext_vector_type(4))) float;
#ifdef DGPPO_STAMPS
__device__ unsigned long long g_dstamps[32];
#define DSTAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_dstamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int32_t dgppo_debug_stamps_dense(unsigned long long* out) {
  return (int32_t)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dstamps), sizeof(unsigned long long) * 32);
}
#else
#define DSTAMP(i)
#endif

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
  const float* mask; int ldm;   // optional [M, N]: Y = (mask > 0) ? result : 0  — the ReLU backward of the layer whose output `mask` is
};


// Stage a [rows x 4*W4] tile of a row-major matrix into LDS (row stride Ls floats).  Loads are issued in batches of 8
// float4 per lane before any LDS store, so 8 global requests per lane are in flight instead of one.
template <bool ALIGNED_LDS>
__device__ inline void stage_tile_f4(const float* __restrict__ src, int ld, int row0, int row_end, int rows, int W4,
                                     float* __restrict__ dst, int Ls, int tid) {
  const int total = rows * W4;
  for (int base = 0; base < total; base += 8 * 256) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = base + u * 256 + tid;
      v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < total) {
        const int r = idx / W4, q = idx - r * W4;
        if (row0 + r < row_end) v[u] = *reinterpret_cast<const float4*>(src + (size_t)(row0 + r) * ld + 4 * q);
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = base + u * 256 + tid;
      if (idx < total) {
        const int r = idx / W4, q = idx - r * W4;
        float* d = dst + r * Ls + 4 * q;
        if (ALIGNED_LDS) *reinterpret_cast<float4*>(d) = v[u];
        else { d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w; }
      }
    }
  }
}

// Persistent workgroups: each keeps ITS W fragments for the whole K in registers (read once from L2), then walks row
// tiles with a stride of gridDim.x.  Per tile: the NEXT tile's X rows are fetched from HBM into registers while the
// current tile (already in LDS) runs on the matrix cores, then written to the other LDS buffer — one LDS-only barrier
// per tile.  Everything in the tile loop is branch-free: K is padded (with zeros, in LDS and in the W fragments) to
// KQ*16, out-of-range rows are clamped on load and masked on store.
//   CG  column groups (waves that split the N/16 column tiles), RG = 4/CG row groups, RTW row tiles (16 rows) per wave,
//   NTW column tiles per wave, KQ = padded K / 16.  Rows per tile RB = 16*RTW*RG.
template <int NTW, int RTW, int CG, int KQ, bool ACC>
__global__ void __launch_bounds__(256) dense_fwd_kernel(DenseArgs a) {
  extern __shared__ float xs_all[];
  constexpr int RG = 4 / CG, RB = 16 * RTW * RG;
  constexpr int KS = KQ * 4;                                   // k-steps of 4 (one MFMA each)
  // LDS row stride = 2 (mod 32) floats: an A-fragment read touches banks (2 li + lq) mod 32, all distinct inside each half of
  // the wave (lq in {0,1} / {2,3}); the odd stride used before put (li, lq = 1) and (li + 1, lq = 0) on the same bank whenever KQ
  // is even (K = 32, 64, 192): SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.5 (profiles/r03_nn_counters.json).  It also makes
  // every float4 slot 8-byte aligned: two ds_write_b64 per slot instead of four ds_write_b32.
  constexpr int Kl = KQ * 16 + 2;
  constexpr int SLOTS = RB * KS;                               // float4 slots per tile
  constexpr int PF = (SLOTS + 255) / 256;                      // prefetch registers (float4) per lane
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lq = lane >> 4;
  const int cg = wave % CG, rg = wave / CG;
  const int K = a.K, N = a.N;
  const bool vec = ((K & 3) == 0) && ((a.ldx & 3) == 0) && ((reinterpret_cast<uintptr_t>(a.X) & 15) == 0);
  const int K4 = K >> 2;
  const int n_tiles = (a.M + RB - 1) / RB;
  DSTAMP(0);
  // ---- W fragments and bias of this wave ----
  float breg[KS][NTW], bias[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    const int col = (cg + CG * t) * 16 + li;
    bias[t] = (a.bias && col < N) ? a.bias[col] : 0.0f;
  }
#pragma unroll
  for (int kk = 0; kk < KS; ++kk) {
    const int k = kk * 4 + lq;
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      const int col = (cg + CG * t) * 16 + li;
      float bv = 0.0f;
      if (k < K && col < N) bv = a.trans_w ? a.W[(size_t)col * a.ldw + k] : a.W[(size_t)k * a.ldw + col];
      breg[kk][t] = bv;
    }
  }
  float4 pf[PF];
  auto fetch = [&](int tile) {          // global -> registers; every load is issued before anything waits
    const int row0 = tile * RB;
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      int idx = u * 256 + tid;
      idx = idx < SLOTS ? idx : SLOTS - 1;                     // duplicates are benign
      const int r = idx / KS, q = idx - r * KS;
      int row = row0 + r;
      row = row < a.M ? row : a.M - 1;
      const int qc = q < K4 ? q : K4 - 1;
      pf[u] = *reinterpret_cast<const float4*>(a.X + (size_t)row * a.ldx + 4 * qc);   // zero padding applied in commit
    }
  };
  auto commit = [&](int b) {            // registers -> LDS buffer b
    float* xs = xs_all + b * (RB * Kl);
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      int idx = u * 256 + tid;
      idx = idx < SLOTS ? idx : SLOTS - 1;
      const int r = idx / KS, q = idx - r * KS;
      float2* d = reinterpret_cast<float2*>(xs + r * Kl + 4 * q);
      const bool pad = q >= K4;                                // touching pf only here keeps the loads in flight
      d[0] = pad ? make_float2(0.0f, 0.0f) : make_float2(pf[u].x, pf[u].y);
      d[1] = pad ? make_float2(0.0f, 0.0f) : make_float2(pf[u].z, pf[u].w);
    }
  };
  auto stage_scalar = [&](int tile, int b) {   // unaligned / K % 4 != 0 fallback (no prefetch)
    float* xs = xs_all + b * (RB * Kl);
    const int row0 = tile * RB;
    for (int idx = tid; idx < RB * (KQ * 16); idx += 256) {
      const int r = idx / (KQ * 16), k = idx - r * (KQ * 16);
      xs[r * Kl + k] = (row0 + r < a.M && k < K) ? a.X[(size_t)(row0 + r) * a.ldx + k] : 0.0f;
    }
  };
  f32x4 acc[RTW][NTW];
  f32x4 yold[ACC ? RTW : 1][ACC ? NTW : 1];
  auto load_yold = [&](int tile) {      // accumulate mode: the old Y tile, issued before the prefetch (in-order vmcnt)
    if constexpr (ACC) {
      const int row0 = tile * RB;
#pragma unroll
      for (int t = 0; t < NTW; ++t) {
        int col = (cg + CG * t) * 16 + li;
        col = col < N ? col : N - 1;
#pragma unroll
        for (int r = 0; r < RTW; ++r)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            int row = row0 + (rg * RTW + r) * 16 + lq * 4 + j;
            row = row < a.M ? row : a.M - 1;
            yold[r][t][j] = a.Y[(size_t)row * a.ldy + col];
          }
      }
    }
  };
  auto compute = [&](int b) {
    const float* xs = xs_all + b * (RB * Kl) + (rg * RTW * 16 + li) * Kl + lq;
#pragma unroll
    for (int r = 0; r < RTW; ++r)
#pragma unroll
      for (int t = 0; t < NTW; ++t) acc[r][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < KS; c += 16) {
      float areg[RTW][16];
#pragma unroll
      for (int r = 0; r < RTW; ++r)
#pragma unroll
        for (int kk = 0; kk < 16; ++kk)
          if (c + kk < KS) areg[r][kk] = xs[r * 16 * Kl + (c + kk) * 4];
      __builtin_amdgcn_sched_barrier(0);          // one LDS round trip for the chunk, then the MFMAs back to back
#pragma unroll
      for (int kk = 0; kk < 16; ++kk)
#pragma unroll
        for (int r = 0; r < RTW; ++r)
#pragma unroll
          for (int t = 0; t < NTW; ++t)
            if (c + kk < KS)
              acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[r][kk], breg[c + kk][t], acc[r][t], 0, 0, 0);
    }
  };
  auto epilogue = [&](int tile) {       // C/D layout: col = lane & 15, row = (lane >> 4) * 4 + reg
    const int row0 = tile * RB;
    // the ReLU-mask values of the whole tile are read BEFORE the first store: read per element in the store loop, every
    // load waits (vmcnt(0)) for the store in front of it as well — one serialised HBM round trip per element
    float mk[RTW][NTW][4];
    if (a.mask != nullptr) {
#pragma unroll
      for (int t = 0; t < NTW; ++t) {
        int col = (cg + CG * t) * 16 + li;
        col = col < N ? col : N - 1;
#pragma unroll
        for (int r = 0; r < RTW; ++r)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            int row = row0 + (rg * RTW + r) * 16 + lq * 4 + j;
            row = row < a.M ? row : a.M - 1;
            mk[r][t][j] = a.mask[(size_t)row * a.ldm + col];
          }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      const int col = (cg + CG * t) * 16 + li;
#pragma unroll
      for (int r = 0; r < RTW; ++r) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int row = row0 + (rg * RTW + r) * 16 + lq * 4 + j;
          float v = acc[r][t][j] + bias[t];
          if constexpr (ACC) v += yold[r][t][j];
          if (a.act == 1) v = fmaxf(v, 0.0f);
          if (a.mask != nullptr) v = (mk[r][t][j] > 0.0f) ? v : 0.0f;
          if (col < N && row < a.M) a.Y[(size_t)row * a.ldy + col] = v;
        }
      }
    }
  };
  // LDS-only barrier: __syncthreads() would also wait for the epilogue's global stores (vmcnt(0)) on every tile.
  // Y is never re-read across waves, so only the ds_writes have to land before the other waves read them.
#define DGPPO_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
  int tile = blockIdx.x;
  if (vec) {
    fetch(tile);
    commit(0);
    __syncthreads();
    DSTAMP(1);
    int cur = 0;
    for (; tile < n_tiles; tile += gridDim.x) {
      int nxt = tile + gridDim.x;
      nxt = nxt < n_tiles ? nxt : n_tiles - 1;                 // last round re-fetches a valid tile; never committed-to-use
      load_yold(tile);
      fetch(nxt);                                              // in flight during the MFMAs below
      __builtin_amdgcn_sched_barrier(0);                       // keep the scheduler from sinking the loads past them
      compute(cur);
      commit(cur ^ 1);                                         // before the stores: vmcnt retires in order
      __builtin_amdgcn_sched_barrier(0);
      epilogue(tile);
      DGPPO_LDS_BARRIER();
      cur ^= 1;
    }
  } else {
    for (; tile < n_tiles; tile += gridDim.x) {
      stage_scalar(tile, 0);
      load_yold(tile);
      __syncthreads();
      compute(0);
      epilogue(tile);
      DGPPO_LDS_BARRIER();
    }
  }
#undef DGPPO_LDS_BARRIER
  DSTAMP(2);
}

template <int NTW, int RTW, int CG, int KQ, bool ACC>
static void launch_dense_acc(const DenseArgs& a, hipStream_t s) {
  constexpr int RB = 16 * RTW * (4 / CG);
  constexpr size_t smem = 2 * (size_t)RB * (KQ * 16 + 2) * sizeof(float);
  static_assert(smem <= 160 * 1024, "dense tile exceeds the 160 KB LDS of a gfx950 CU");
  // persistent grid = what is actually resident (registers and LDS both limit it); queried once per instantiation
  static thread_local int cap = 0;
  if (cap == 0) {
    int per_cu = 0, dev = 0, cus = 256;
    if (smem > 64 * 1024)                          // above the default dynamic-LDS limit: opt in explicitly
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_fwd_kernel<NTW, RTW, CG, KQ, ACC>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(
            &dense_fwd_kernel<NTW, RTW, CG, KQ, ACC>), 256, smem) != hipSuccess || per_cu < 1) per_cu = 1;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
    cap = per_cu * cus;
  }
  const int n_tiles = cdiv(a.M, RB);
  const int grid = n_tiles < cap ? n_tiles : cap;
  hipLaunchKernelGGL((dense_fwd_kernel<NTW, RTW, CG, KQ, ACC>), dim3(grid), dim3(256), smem, s, a);
}

template <int NTW, int RTW, int CG, int KQ>
static void launch_dense_kq(const DenseArgs& a, hipStream_t s) {
  if (a.accumulate) launch_dense_acc<NTW, RTW, CG, KQ, true>(a, s);
  else launch_dense_acc<NTW, RTW, CG, KQ, false>(a, s);
}

template <int NTW, int RTW, int CG>
static void launch_dense(const DenseArgs& a, hipStream_t s) {
  const int kq = cdiv(a.K, 16);                    // padded-K instantiations: 16,32,48,64,96,128,144,192,256
  if (kq <= 1) launch_dense_kq<NTW, RTW, CG, 1>(a, s);
  else if (kq == 2) launch_dense_kq<NTW, RTW, CG, 2>(a, s);
  else if (kq == 3) launch_dense_kq<NTW, RTW, CG, 3>(a, s);
  else if (kq == 4) launch_dense_kq<NTW, RTW, CG, 4>(a, s);
  else if (kq <= 6) launch_dense_kq<NTW, RTW, CG, 6>(a, s);
  else if (kq <= 8) launch_dense_kq<NTW, RTW, CG, 8>(a, s);
  else if (kq == 9) launch_dense_kq<NTW, RTW, CG, 9>(a, s);
  else if (kq <= 12) launch_dense_kq<NTW, RTW, CG, 12>(a, s);
  else launch_dense_kq<NTW, RTW, CG, 16>(a, s);
}

// ---- narrow inputs (K <= 16): the query projection of the first GNN layer (K = 8), the input gradients of the 4- /
// 1- / 2-wide heads.  These move 100+ bytes per row for a handful of FMAs: a tile pipeline with a barrier per 32 rows
// (above) runs at its latency floor (23 us for 17 MB).  Here a thread owns four consecutive output columns of a row, the
// whole W (<= 16 x 192) sits in LDS in [k][n] order (transposed on the way in if trans_w), X is read straight from global
// memory (the threads of a row read the same addresses: one L1 broadcast) and Y is written as coalesced 16-byte stores.
#define DSK_MAXK 16
template <bool VECY>
__global__ void __launch_bounds__(256) dense_smallk_kernel(DenseArgs a) {
  __shared__ float sW[DSK_MAXK * 192 + 192];
  const int K = a.K, N = a.N, nq = (N + 3) >> 2, Nl = nq * 4;
  for (int idx = threadIdx.x; idx < K * Nl; idx += 256) {
    const int k = idx / Nl, c = idx - k * Nl;
    float w = 0.0f;
    if (c < N) w = a.trans_w ? a.W[(size_t)c * a.ldw + k] : a.W[(size_t)k * a.ldw + c];
    sW[idx] = w;
  }
  for (int c = threadIdx.x; c < Nl; c += 256) sW[K * Nl + c] = (a.bias && c < N) ? a.bias[c] : 0.0f;
  __syncthreads();
  const long items = (long)a.M * nq;
  const uint32_t rcp = (uint32_t)((0x100000000ull + nq - 1) / nq);    // item / nq as a multiply-high (exact below 2^32 / nq items)
  for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < items; it += (long)gridDim.x * 256) {
    const int row = (int)__umulhi((uint32_t)it, rcp);
    const int c0 = ((int)it - row * nq) * 4;
    const float* x = a.X + (size_t)row * a.ldx;
    float4 acc = *reinterpret_cast<const float4*>(sW + K * Nl + c0);
    for (int k = 0; k < K; ++k) {
      const float xv = x[k];
      const float4 w = *reinterpret_cast<const float4*>(sW + k * Nl + c0);
      acc.x = fmaf(xv, w.x, acc.x); acc.y = fmaf(xv, w.y, acc.y); acc.z = fmaf(xv, w.z, acc.z); acc.w = fmaf(xv, w.w, acc.w);
    }
    float* y = a.Y + (size_t)row * a.ldy + c0;
    float r[4] = {acc.x, acc.y, acc.z, acc.w};
    if (VECY) {
      if (a.accumulate) { const float4 o = *reinterpret_cast<const float4*>(y); r[0] += o.x; r[1] += o.y; r[2] += o.z; r[3] += o.w; }
      if (a.act == 1) { r[0] = fmaxf(r[0], 0.f); r[1] = fmaxf(r[1], 0.f); r[2] = fmaxf(r[2], 0.f); r[3] = fmaxf(r[3], 0.f); }
      if (a.mask) {
        const float4 mk = *reinterpret_cast<const float4*>(a.mask + (size_t)row * a.ldm + c0);
        r[0] = mk.x > 0.f ? r[0] : 0.f; r[1] = mk.y > 0.f ? r[1] : 0.f; r[2] = mk.z > 0.f ? r[2] : 0.f; r[3] = mk.w > 0.f ? r[3] : 0.f;
      }
      *reinterpret_cast<float4*>(y) = make_float4(r[0], r[1], r[2], r[3]);
    } else {
      for (int u = 0; u < 4; ++u) {
        if (c0 + u >= N) break;
        float v = r[u];
        if (a.accumulate) v += y[u];
        if (a.act == 1) v = fmaxf(v, 0.f);
        if (a.mask) v = a.mask[(size_t)row * a.ldm + c0 + u] > 0.f ? v : 0.f;
        y[u] = v;
      }
    }
  }
}

static bool dense_smallk_ok(const DenseArgs& a) {
  const long nq = (a.N + 3) / 4;                    // the multiply-high row index is exact while items * nq < 2^32
  return a.K <= DSK_MAXK && a.N <= 192 && (long)a.M * nq * nq < (1l << 32) && !getenv("DGPPO_DENSE_NO_SMALLK") &&
         !getenv("DGPPO_DENSE_NO_SMALLK_FWD");
}
static void launch_dense_smallk(const DenseArgs& a, hipStream_t s) {
  const long items = (long)a.M * ((a.N + 3) / 4);
  long blocks = (items + 256 * 4 - 1) / (256 * 4);            // >= 4 items per thread amortise the staging of W
  blocks = blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks);
  const bool vec = (a.N & 3) == 0 && (a.ldy & 3) == 0 && (reinterpret_cast<uintptr_t>(a.Y) & 15) == 0 &&
                   (!a.mask || ((a.ldm & 3) == 0 && (reinterpret_cast<uintptr_t>(a.mask) & 15) == 0));
  if (vec) hipLaunchKernelGGL(dense_smallk_kernel<true>, dim3((int)blocks), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(dense_smallk_kernel<false>, dim3((int)blocks), dim3(256), 0, s, a);
}

int32_t dense_fwd_launch(const DenseArgs& a, hipStream_t s) {
  DGPPO_REQUIRE(a.M >= 0 && a.K >= 1 && a.N >= 1, "dense: bad shape M=%d K=%d N=%d", a.M, a.K, a.N);
  DGPPO_REQUIRE(a.N <= 192 && a.K <= 256, "dense: N <= 192 and K <= 256 supported (N=%d K=%d)", a.N, a.K);
  DGPPO_REQUIRE(a.X && a.W && a.Y, "dense: NULL operand");
  DGPPO_REQUIRE(a.ldx >= a.K && a.ldy >= a.N, "dense: leading dimensions too small");
  if (a.M == 0) return 0;
  const int nt = cdiv(a.N, 16);
  if (dense_smallk_ok(a)) launch_dense_smallk(a, s);
  else if (nt >= 9) launch_dense<3, 2, 4>(a, s);   // N in (128, 192]
  else if (nt >= 5) launch_dense<2, 2, 4>(a, s);   // N in (64, 128]
  else if (nt == 4) launch_dense<1, 2, 4>(a, s);   // N in (48, 64]
  else if (nt == 3) launch_dense<2, 1, 2>(a, s);   // N in (32, 48]
  else if (nt == 2) launch_dense<1, 1, 2>(a, s);   // N in (16, 32]
  else launch_dense<1, 1, 1>(a, s);                // N <= 16  (64-row tiles, one row tile per wave)
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
  float* part;        // scratch for per-workgroup partials [grid][part_stride], or NULL: atomicAdd straight into dW/db
  int part_stride;    // >= K*N + N
  size_t ws_bytes;    // size of the caller's scratch buffer behind `part`
};

// A workgroup owns a contiguous chunk of rows and walks it in tiles of 32.  X[32,K] and dY[32,N] tiles live in LDS
// (double-buffered); the NEXT tile's rows are fetched from HBM into registers while the current one runs on the matrix
// cores and are written to the other buffer afterwards — one LDS-only barrier per tile (no global stores in the loop, and
// a __syncthreads() would wait for the prefetch).  Wave w accumulates the output row-tiles kt = w, w+4, ... x all NT
// column tiles over the tile's 8 four-row slabs (A[i=k][kk=m] = X[m][k], B[kk=m][j] = dY[m][j]).  K and N are padded to
// KT*16 / NT*16 with zeros in LDS so the loop is branch-free; db's column sums are accumulated per lane from the
// prefetch registers and reduced once at the end.  Write-out: every workgroup stores its K*N (+N) partial sums to a
// slab of the caller's workspace and dense_bwd_w_reduce_kernel folds the slabs into dW/db — with hundreds of workgroups
// finishing together, atomicAdd on the same K*N addresses serialises (measured: 137 us vs 10 us for K=64, N=4).  Only
// launches of <= 4 workgroups (or a NULL / too small workspace) use atomicAdd directly.
#define BW_ROWS 32
template <int NT, int KT>
__global__ void __launch_bounds__(256) dense_bwd_w_kernel(DenseBwdWArgs a) {
  extern __shared__ float sm[];
  constexpr int KTW = (KT + 3) / 4;
  // row strides = 16 mod 32 floats: the four slab rows of a fragment read hit disjoint bank halves
  constexpr int Kl = (KT & 1) ? KT * 16 : KT * 16 + 16;
  constexpr int Nl = (NT & 1) ? NT * 16 : NT * 16 + 16;
  constexpr int BUF = BW_ROWS * (Kl + Nl);
  constexpr int XQ = KT * 4, YQ = NT * 4;                      // float4 slots per padded row
  constexpr int PFX = (BW_ROWS * XQ + 255) / 256, PFY = (BW_ROWS * YQ + 255) / 256;
  const int K = a.K, N = a.N;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lq = lane >> 4;
  const int m_begin = blockIdx.x * a.rows_per_block;
  const int m_end = min(a.M, m_begin + a.rows_per_block);
  const bool vec = ((K & 3) == 0) && ((a.ldx & 3) == 0) && ((reinterpret_cast<uintptr_t>(a.X) & 15) == 0) &&
                   ((N & 3) == 0) && ((a.ldy & 3) == 0) && ((reinterpret_cast<uintptr_t>(a.dY) & 15) == 0);
  const int K4 = K >> 2, N4 = N >> 2;
  f32x4 acc[KTW][NT];
#pragma unroll
  for (int i = 0; i < KTW; ++i)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float4 pfx[PFX], pfy[PFY], csum[PFY];
#pragma unroll
  for (int u = 0; u < PFY; ++u) csum[u] = make_float4(0.f, 0.f, 0.f, 0.f);

  auto fetch = [&](int m0) {            // global -> registers (rows clamped; masking happens in commit)
#pragma unroll
    for (int u = 0; u < PFX; ++u) {
      int idx = u * 256 + tid;
      idx = idx < BW_ROWS * XQ ? idx : BW_ROWS * XQ - 1;
      const int r = idx / XQ, q = idx - r * XQ;
      int row = m0 + r;
      row = row < a.M ? row : a.M - 1;
      pfx[u] = *reinterpret_cast<const float4*>(a.X + (size_t)row * a.ldx + 4 * (q < K4 ? q : K4 - 1));
    }
#pragma unroll
    for (int u = 0; u < PFY; ++u) {
      int idx = u * 256 + tid;
      idx = idx < BW_ROWS * YQ ? idx : BW_ROWS * YQ - 1;
      const int r = idx / YQ, q = idx - r * YQ;
      int row = m0 + r;
      row = row < a.M ? row : a.M - 1;
      pfy[u] = *reinterpret_cast<const float4*>(a.dY + (size_t)row * a.ldy + 4 * (q < N4 ? q : N4 - 1));
    }
  };
  auto commit = [&](int m0, int b) {    // registers -> LDS buffer b; rows >= m_end and padded columns become zeros
    float* xs = sm + b * BUF;
    float* ys = xs + BW_ROWS * Kl;
#pragma unroll
    for (int u = 0; u < PFX; ++u) {
      const int idx = u * 256 + tid;
      const int r = idx / XQ, q = idx - r * XQ;
      const bool ok = (m0 + r < m_end) && (q < K4);
      const float4 v = ok ? pfx[u] : make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < BW_ROWS * XQ) *reinterpret_cast<float4*>(xs + r * Kl + 4 * q) = v;
    }
#pragma unroll
    for (int u = 0; u < PFY; ++u) {
      const int idx = u * 256 + tid;
      const int r = idx / YQ, q = idx - r * YQ;
      const bool ok = (m0 + r < m_end) && (q < N4) && (idx < BW_ROWS * YQ);
      const float4 v = ok ? pfy[u] : make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < BW_ROWS * YQ) *reinterpret_cast<float4*>(ys + r * Nl + 4 * q) = v;
      csum[u].x += v.x; csum[u].y += v.y; csum[u].z += v.z; csum[u].w += v.w;
    }
  };
  auto stage_scalar = [&](int m0) {     // unaligned / odd-width fallback: synchronous, buffer 0
    float* xs = sm;
    float* ys = xs + BW_ROWS * Kl;
    for (int idx = tid; idx < BW_ROWS * KT * 16; idx += 256) {
      const int r = idx / (KT * 16), k = idx - r * (KT * 16);
      xs[r * Kl + k] = (m0 + r < m_end && k < K) ? a.X[(size_t)(m0 + r) * a.ldx + k] : 0.0f;
    }
    for (int idx = tid; idx < BW_ROWS * NT * 16; idx += 256) {
      const int r = idx / (NT * 16), c = idx - r * (NT * 16);
      ys[r * Nl + c] = (m0 + r < m_end && c < N) ? a.dY[(size_t)(m0 + r) * a.ldy + c] : 0.0f;
    }
  };
  auto compute = [&](int b) {
    const float* xs = sm + b * BUF + lq * Kl + li;
    const float* ys = sm + b * BUF + BW_ROWS * Kl + lq * Nl + li;
#pragma unroll
    for (int s4 = 0; s4 < BW_ROWS / 4; ++s4) {
      float bv[NT], av[KTW];
#pragma unroll
      for (int t = 0; t < NT; ++t) bv[t] = ys[s4 * 4 * Nl + t * 16];
#pragma unroll
      for (int i = 0; i < KTW; ++i) {
        int kt = wave + 4 * i;
        kt = kt < KT ? kt : KT - 1;                            // surplus waves redo the last tile (never written back)
        av[i] = xs[s4 * 4 * Kl + kt * 16];
      }
#pragma unroll
      for (int i = 0; i < KTW; ++i)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[t], acc[i][t], 0, 0, 0);
    }
  };
#define DGPPO_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
  float colsum = 0.0f;                  // scalar path only
  if (vec) {
    fetch(m_begin);
    commit(m_begin, 0);
    __syncthreads();
    int cur = 0;
    for (int m0 = m_begin; m0 < m_end; m0 += BW_ROWS) {
      fetch(m0 + BW_ROWS);                                     // past m_end on the last round: masked to zeros in commit
      __builtin_amdgcn_sched_barrier(0);
      compute(cur);
      commit(m0 + BW_ROWS, cur ^ 1);
      DGPPO_LDS_BARRIER();
      cur ^= 1;
    }
  } else {
    for (int m0 = m_begin; m0 < m_end; m0 += BW_ROWS) {
      stage_scalar(m0);
      __syncthreads();
      if (a.db != nullptr && tid < N) {
        const float* ys = sm + BW_ROWS * Kl;
        float cs = 0.0f;
#pragma unroll 8
        for (int r = 0; r < BW_ROWS; ++r) cs += ys[r * Nl + tid];
        colsum += cs;
      }
      compute(0);
      __syncthreads();
    }
  }
#undef DGPPO_LDS_BARRIER
  float* slab = a.part ? a.part + (size_t)blockIdx.x * a.part_stride : nullptr;
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
        if (krow >= K) continue;
        if (slab) slab[krow * N + col] = acc[i][t][r];
        else atomicAdd(a.dW + (size_t)krow * a.ldw + col, acc[i][t][r]);
      }
    }
  }
  if (a.db != nullptr) {
    float cs = colsum;
    if (vec) {                          // reduce the per-lane column sums through LDS: red[row slot][column]
      __syncthreads();
      float* red = sm;
#pragma unroll
      for (int u = 0; u < PFY; ++u) {
        const int idx = u * 256 + tid;
        const int r = idx / YQ, q = idx - r * YQ;
        if (idx < BW_ROWS * YQ) *reinterpret_cast<float4*>(red + r * (YQ * 4) + 4 * q) = csum[u];
      }
      __syncthreads();
      cs = 0.0f;
      if (tid < N) {
#pragma unroll 8
        for (int r = 0; r < BW_ROWS; ++r) cs += red[r * (YQ * 4) + tid];
      }
    }
    if (tid < N) {
      if (slab) slab[K * N + tid] = cs;
      else atomicAdd(a.db + tid, cs);
    }
  }
}

// Second stage: dW[k][n] += sum_g part[g][k*N + n], db[n] += sum_g part[g][K*N + n].  blockIdx.y splits the slabs so
// that small outputs still fill the device; the splits meet in at most gridDim.y atomicAdds per element.
__global__ void __launch_bounds__(256) dense_bwd_w_reduce_kernel(const float* __restrict__ part, int part_stride, int G,
                                                                 float* __restrict__ dW, int ldw, float* __restrict__ db,
                                                                 int K, int N) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  const int E = K * N + (db ? N : 0);
  if (e >= E) return;
  const int per = (G + gridDim.y - 1) / gridDim.y;
  const int g0 = blockIdx.y * per, g1 = min(G, g0 + per);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int g = g0;
  for (; g + 3 < g1; g += 4) {
    s0 += part[(size_t)g * part_stride + e];
    s1 += part[(size_t)(g + 1) * part_stride + e];
    s2 += part[(size_t)(g + 2) * part_stride + e];
    s3 += part[(size_t)(g + 3) * part_stride + e];
  }
  for (; g < g1; ++g) s0 += part[(size_t)g * part_stride + e];
  const float sum = (s0 + s1) + (s2 + s3);
  if (g0 >= g1) return;
  if (e < K * N) {
    const int k = e / N, n = e - k * N;
    atomicAdd(dW + (size_t)k * ldw + n, sum);
  } else {
    atomicAdd(db + (e - K * N), sum);
  }
}

// Deferred second stage: dgppo_dense_bwd_w_deferred launches only the partial-slab kernel and hands the description of the
// pending reduction back to the caller, who flushes up to DGPPO_REDUCE_BATCH of them with ONE launch
// (dgppo_dense_bwd_w_reduce_batch; blockIdx.z = reduction).  A backward pass has ~12 weight gradients: 12 reduce launches
// of ~6 us become one.
static thread_local dgppo_reduce_desc* g_defer = nullptr;

struct ReduceBatch {
  int n;
  dgppo_reduce_desc d[DGPPO_REDUCE_BATCH];
};

__global__ void __launch_bounds__(256) dense_bwd_w_reduce_batch_kernel(ReduceBatch b) {
  const dgppo_reduce_desc& d = b.d[blockIdx.z];
  const int K = d.K, N = d.N, G = d.slabs;
  const int e = blockIdx.x * 256 + threadIdx.x;
  const int E = K * N + (d.db ? N : 0);
  if (e >= E) return;
  // the same split of the slabs over blockIdx.y as the single reduce kernel: splits meet in <= gridDim.y atomicAdds
  const int splits = min((int)gridDim.y, max(1, (G + 31) / 32));
  if ((int)blockIdx.y >= splits) return;
  const int per = (G + splits - 1) / splits;
  const int g0 = blockIdx.y * per, g1 = min(G, g0 + per);
  if (g0 >= g1) return;
  const float* part = d.part;
  const size_t st = (size_t)d.part_stride;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int g = g0;
  for (; g + 3 < g1; g += 4) {
    s0 += part[(size_t)g * st + e]; s1 += part[(size_t)(g + 1) * st + e];
    s2 += part[(size_t)(g + 2) * st + e]; s3 += part[(size_t)(g + 3) * st + e];
  }
  for (; g < g1; ++g) s0 += part[(size_t)g * st + e];
  const float sum = (s0 + s1) + (s2 + s3);
  if (e < K * N) {
    const int k = e / N, n = e - k * N;
    atomicAdd(d.dW + (size_t)k * d.ldw + n, sum);
  } else {
    atomicAdd(d.db + (e - K * N), sum);
  }
}

// second stage of one weight gradient: launched now, or recorded for a batched launch
static void reduce_or_defer(hipStream_t s, const float* part, int stride, int grid, float* dW, int ldw, float* db, int K, int N) {
  if (g_defer != nullptr) {
    dgppo_reduce_desc& d = *g_defer;
    d.part = part; d.part_stride = stride; d.slabs = grid; d.dW = dW; d.ldw = ldw; d.db = db; d.K = K; d.N = N; d.pending = 1;
    return;
  }
  const int E = K * N + (db ? N : 0);
  int splits = cdiv(grid, 32);
  splits = splits < 1 ? 1 : (splits > 64 ? 64 : splits);
  hipLaunchKernelGGL(dense_bwd_w_reduce_kernel, dim3(cdiv(E, 256), splits), dim3(256), 0, s, part, stride, grid, dW, ldw, db, K, N);
}

template <int NT, int KT>
static void launch_bwd_w_kt(DenseBwdWArgs a, hipStream_t s) {
  constexpr int Kl = (KT & 1) ? KT * 16 : KT * 16 + 16;
  constexpr int Nl = (NT & 1) ? NT * 16 : NT * 16 + 16;
  constexpr size_t smem = 2 * sizeof(float) * BW_ROWS * (Kl + Nl);
  static_assert(smem <= 160 * 1024, "dense_bwd_w tiles exceed the 160 KB LDS of a gfx950 CU");
  static thread_local int cap = 0;      // resident workgroups on the device, queried once per instantiation
  if (cap == 0) {
    int per_cu = 0, dev = 0, cus = 256;
    if (smem > 64 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_bwd_w_kernel<NT, KT>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(
            &dense_bwd_w_kernel<NT, KT>), 256, smem) != hipSuccess || per_cu < 1) per_cu = 1;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
    cap = per_cu * cus;
  }
  // one resident wave of workgroups; rows per workgroup a multiple of the 32-row tile, at least two tiles
  int rpb = cdiv(a.M, cap);
  rpb = ((rpb + BW_ROWS - 1) / BW_ROWS) * BW_ROWS;
  if (rpb < 2 * BW_ROWS) rpb = 2 * BW_ROWS;
  // partial slabs: stride rounded to 64 floats; shrink the grid if the scratch buffer cannot hold one slab per workgroup
  const int stride = ((a.K * a.N + a.N + 63) / 64) * 64;
  const size_t ws_bytes = a.ws_bytes;
  float* ws = cdiv(a.M, rpb) > 4 ? a.part : nullptr;
  if (ws) {
    const long max_slabs = (long)(ws_bytes / (sizeof(float) * stride));
    if (max_slabs < 8) ws = nullptr;
    else if (cdiv(a.M, rpb) > max_slabs) rpb = ((cdiv(a.M, (int)max_slabs) + BW_ROWS - 1) / BW_ROWS) * BW_ROWS;
  }
  a.rows_per_block = rpb;
  a.part = ws;
  a.part_stride = stride;
  const int grid = cdiv(a.M, rpb);
  hipLaunchKernelGGL((dense_bwd_w_kernel<NT, KT>), dim3(grid), dim3(256), smem, s, a);
  if (ws) reduce_or_defer(s, ws, stride, grid, a.dW, a.ldw, a.db, a.K, a.N);
}

template <int NT>
static void launch_bwd_w(const DenseBwdWArgs& a, hipStream_t s) {
  const int kt = cdiv(a.K, 16);                    // padded-K instantiations: 16,32,48,64,96,128,144,192,256
  if (kt <= 1) launch_bwd_w_kt<NT, 1>(a, s);
  else if (kt == 2) launch_bwd_w_kt<NT, 2>(a, s);
  else if (kt == 3) launch_bwd_w_kt<NT, 3>(a, s);
  else if (kt == 4) launch_bwd_w_kt<NT, 4>(a, s);
  else if (kt <= 6) launch_bwd_w_kt<NT, 6>(a, s);
  else if (kt <= 8) launch_bwd_w_kt<NT, 8>(a, s);
  else if (kt == 9) launch_bwd_w_kt<NT, 9>(a, s);
  else if (kt <= 12) launch_bwd_w_kt<NT, 12>(a, s);
  else launch_bwd_w_kt<NT, 16>(a, s);
}

// ---- narrow weight gradients (K <= 16, N <= 64: dMcat of the first GNN layer, K = 8, N = 24) ------------------------
// A thread owns four output columns and all K rows of dW (4K accumulators) and walks its share of the M input rows: dY is
// read as coalesced 16-byte loads, X[m][0..K) is the same address for the N/4 threads of a row.  The row-lanes of a
// workgroup meet in LDS (float atomics, once per workgroup), the workgroup writes one partial slab, and the common
// second stage (dense_bwd_w_reduce_kernel) folds the slabs into dW / db.
template <int KMAX, bool VECX>
__global__ void __launch_bounds__(256) dense_bwd_w_smallk_kernel(DenseBwdWArgs a) {
  extern __shared__ float sP[];                     // [256][K*4 + 4 + 1]: every thread's accumulators, then summed per column
  const int K = a.K, N = a.N, nq = N >> 2;
  const int per = K * 4 + 4, Pl = per + 1;
  const int lanes = 256 / nq;                       // row-lanes of the workgroup
  const int rl = threadIdx.x / nq, cq = threadIdx.x - rl * nq;
  float4 acc[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 accb = make_float4(0.f, 0.f, 0.f, 0.f);
  if (rl < lanes) {
    const int r0 = blockIdx.x * a.rows_per_block;
    const int r1 = min(a.M, r0 + a.rows_per_block);
    // UR rows in flight per thread: the loop is a chain of dependent loads otherwise (one row = 1 + K/4 requests)
    constexpr int UR = 4;
    for (int rb = r0 + rl; rb < r1; rb += lanes * UR) {
      float4 d[UR];
      float xv[UR][KMAX];
#pragma unroll
      for (int u = 0; u < UR; ++u) {
        const int r = rb + u * lanes;
        const bool in = r < r1;
        const int rr = in ? r : r0;
        d[u] = *reinterpret_cast<const float4*>(a.dY + (size_t)rr * a.ldy + cq * 4);
        if (!in) d[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        const float* x = a.X + (size_t)rr * a.ldx;
        if (VECX) {
#pragma unroll
          for (int k4 = 0; k4 < KMAX / 4; ++k4) {
            if (k4 * 4 < K) {
              const float4 t = *reinterpret_cast<const float4*>(x + k4 * 4);
              xv[u][k4 * 4] = t.x; xv[u][k4 * 4 + 1] = t.y; xv[u][k4 * 4 + 2] = t.z; xv[u][k4 * 4 + 3] = t.w;
            }
          }
        } else {
#pragma unroll
          for (int k = 0; k < KMAX; ++k) if (k < K) xv[u][k] = x[k];
        }
      }
#pragma unroll
      for (int u = 0; u < UR; ++u) {
        accb.x += d[u].x; accb.y += d[u].y; accb.z += d[u].z; accb.w += d[u].w;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
          if (k < K) {
            acc[k].x = fmaf(xv[u][k], d[u].x, acc[k].x); acc[k].y = fmaf(xv[u][k], d[u].y, acc[k].y);
            acc[k].z = fmaf(xv[u][k], d[u].z, acc[k].z); acc[k].w = fmaf(xv[u][k], d[u].w, acc[k].w);
          }
        }
      }
    }
  }
  float* mine = sP + threadIdx.x * Pl;
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    if (k < K) { mine[k * 4] = acc[k].x; mine[k * 4 + 1] = acc[k].y; mine[k * 4 + 2] = acc[k].z; mine[k * 4 + 3] = acc[k].w; }
  }
  mine[K * 4] = accb.x; mine[K * 4 + 1] = accb.y; mine[K * 4 + 2] = accb.z; mine[K * 4 + 3] = accb.w;
  __syncthreads();
  // output element e = k*N + c (k == K: the bias row) is the sum over the row-lanes of thread (rl, cq = c/4), slot k*4 + c%4
  float* out = a.part + (size_t)blockIdx.x * a.part_stride;
  const int E = K * N + N;
  for (int e = threadIdx.x; e < E; e += 256) {
    const int k = e / N, c = e - k * N;
    const float* src = sP + (c >> 2) * Pl + k * 4 + (c & 3);
    float s0 = 0.f, s1 = 0.f;
    int l = 0;
    for (; l + 1 < lanes; l += 2) { s0 += src[(size_t)l * nq * Pl]; s1 += src[(size_t)(l + 1) * nq * Pl]; }
    if (l < lanes) s0 += src[(size_t)l * nq * Pl];
    out[e] = s0 + s1;
  }
}

static bool launch_bwd_w_smallk(DenseBwdWArgs a, hipStream_t s) {
  if (a.K > 16 || a.N > 64 || (a.N & 3) || (a.ldy & 3) || (reinterpret_cast<uintptr_t>(a.dY) & 15) || a.M < 4096 ||
      getenv("DGPPO_DENSE_NO_SMALLK") || getenv("DGPPO_DENSE_NO_SMALLK_BWD"))
    return false;
  const int stride = ((a.K * a.N + a.N + 63) / 64) * 64;
  const long max_slabs = a.part ? (long)(a.ws_bytes / (sizeof(float) * stride)) : 0;
  if (max_slabs < 8) return false;
  int grid = a.M >= (1 << 19) ? 1024 : 512;          // 2-4 workgroups per CU
  if (grid > max_slabs) grid = (int)max_slabs;
  int rpb = cdiv(a.M, grid);
  grid = cdiv(a.M, rpb);
  a.rows_per_block = rpb;
  a.part_stride = stride;
  const size_t smem = sizeof(float) * 256 * (a.K * 4 + 5);            // <= 69 KB
  const bool vx = (a.K & 3) == 0 && (a.ldx & 3) == 0 && (reinterpret_cast<uintptr_t>(a.X) & 15) == 0;
  if (a.K <= 8) {
    if (vx) hipLaunchKernelGGL((dense_bwd_w_smallk_kernel<8, true>), dim3(grid), dim3(256), smem, s, a);
    else hipLaunchKernelGGL((dense_bwd_w_smallk_kernel<8, false>), dim3(grid), dim3(256), smem, s, a);
  } else {
    static thread_local bool opted = false;
    if (!opted) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_bwd_w_smallk_kernel<16, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 256 * 69 * 4);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_bwd_w_smallk_kernel<16, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 256 * 69 * 4);
      opted = true;
    }
    if (vx) hipLaunchKernelGGL((dense_bwd_w_smallk_kernel<16, true>), dim3(grid), dim3(256), smem, s, a);
    else hipLaunchKernelGGL((dense_bwd_w_smallk_kernel<16, false>), dim3(grid), dim3(256), smem, s, a);
  }
  reduce_or_defer(s, a.part, stride, grid, a.dW, a.ldw, a.db, a.K, a.N);
  return true;
}

int32_t dense_bwd_w_launch(DenseBwdWArgs a, hipStream_t s) {
  DGPPO_REQUIRE(a.M >= 0 && a.K >= 1 && a.N >= 1, "dense_bwd_w: bad shape");
  DGPPO_REQUIRE(a.N <= 192 && a.K <= 256, "dense_bwd_w: N <= 192 and K <= 256 supported (N=%d K=%d)", a.N, a.K);
  DGPPO_REQUIRE(a.X && a.dY && a.dW, "dense_bwd_w: NULL operand");
  DGPPO_REQUIRE(a.ldx >= a.K && a.ldy >= a.N && a.ldw >= a.N, "dense_bwd_w: leading dimensions too small");
  if (a.M == 0) return 0;
  const int nt = cdiv(a.N, 16);
  if (launch_bwd_w_smallk(a, s)) {}
  else if (nt <= 1) launch_bwd_w<1>(a, s);
  else if (nt <= 2) launch_bwd_w<2>(a, s);
  else if (nt <= 4) launch_bwd_w<4>(a, s);
  else if (nt <= 6) launch_bwd_w<6>(a, s);
  else launch_bwd_w<12>(a, s);
  DGPPO_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t dgppo_dense_fwd(const float* X, int32_t ldx, const float* W, int32_t ldw, const float* bias, float* Y,
                                   int32_t ldy, int32_t M, int32_t K, int32_t N, int32_t act, int32_t accumulate,
                                   int32_t trans_w, const float* relu_mask, int32_t ldm, void* stream) {
  DGPPO_REQUIRE(relu_mask == nullptr || ldm >= N, "dense: mask leading dimension too small");
  DenseArgs a{X, ldx, W, ldw, bias, Y, ldy, M, K, N, act, accumulate, trans_w, relu_mask, ldm};
  return dense_fwd_launch(a, (hipStream_t)stream);
}

extern "C" int32_t dgppo_dense_bwd_w(const float* X, int32_t ldx, const float* dY, int32_t ldy, float* dW, int32_t ldw,
                                     float* db, int32_t M, int32_t K, int32_t N, float* workspace, int64_t workspace_bytes,
                                     void* stream) {
  DGPPO_REQUIRE(workspace_bytes >= 0 && (workspace != nullptr || workspace_bytes == 0), "dense_bwd_w: bad workspace");
  DGPPO_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15) == 0, "dense_bwd_w: workspace must be 16-byte aligned");
  DenseBwdWArgs a{X, ldx, dY, ldy, dW, ldw, db, M, K, N, 0, workspace, 0, (size_t)workspace_bytes};
  return dense_bwd_w_launch(a, (hipStream_t)stream);
}

extern "C" int32_t dgppo_dense_bwd_w_deferred(const float* X, int32_t ldx, const float* dY, int32_t ldy, float* dW,
                                              int32_t ldw, float* db, int32_t M, int32_t K, int32_t N, float* workspace,
                                              int64_t workspace_bytes, dgppo_reduce_desc* pending, void* stream) {
  DGPPO_REQUIRE(pending != nullptr, "dense_bwd_w_deferred: pending is NULL");
  DGPPO_REQUIRE(workspace_bytes >= 0 && (workspace != nullptr || workspace_bytes == 0), "dense_bwd_w: bad workspace");
  DGPPO_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15) == 0, "dense_bwd_w: workspace must be 16-byte aligned");
  pending->pending = 0;
  DenseBwdWArgs a{X, ldx, dY, ldy, dW, ldw, db, M, K, N, 0, workspace, 0, (size_t)workspace_bytes};
  g_defer = pending;
  const int32_t rc = dense_bwd_w_launch(a, (hipStream_t)stream);
  g_defer = nullptr;
  return rc;
}

extern "C" int32_t dgppo_dense_bwd_w_reduce_batch(const dgppo_reduce_desc* descs, int32_t n, void* stream) {
  DGPPO_REQUIRE(n >= 0 && (n == 0 || descs != nullptr), "dense_bwd_w_reduce_batch: bad arguments");
  for (int32_t i0 = 0; i0 < n; i0 += DGPPO_REDUCE_BATCH) {
    ReduceBatch b;
    b.n = 0;
    int emax = 0, gmax = 0;
    for (int32_t i = i0; i < n && b.n < DGPPO_REDUCE_BATCH; ++i) {
      if (!descs[i].pending) continue;
      DGPPO_REQUIRE(descs[i].part && descs[i].dW && descs[i].slabs >= 1 && descs[i].K >= 1 && descs[i].N >= 1,
                    "dense_bwd_w_reduce_batch: bad descriptor %d", i);
      b.d[b.n++] = descs[i];
      const int E = descs[i].K * descs[i].N + (descs[i].db ? descs[i].N : 0);
      emax = E > emax ? E : emax;
      gmax = descs[i].slabs > gmax ? descs[i].slabs : gmax;
    }
    if (b.n == 0) continue;
    int splits = cdiv(gmax, 32);
    splits = splits < 1 ? 1 : (splits > 64 ? 64 : splits);
    hipLaunchKernelGGL(dense_bwd_w_reduce_batch_kernel, dim3(cdiv(emax, 256), splits, b.n), dim3(256), 0, (hipStream_t)stream, b);
  }
  DGPPO_LAUNCH_CHECK();
  return 0;
}

// one slab of K*N + N floats (rounded to 64) per resident workgroup; 1024 workgroups cover every instantiation's
// residency (<= 4 per CU x 256 CUs)
extern "C" int64_t dgppo_dense_bwd_w_workspace_bytes(int32_t K, int32_t N) {
  if (K < 1 || N < 1) return 0;
  const int64_t stride = (((int64_t)K * N + N + 63) / 64) * 64;
  return stride * 1024 * (int64_t)sizeof(float);
}
