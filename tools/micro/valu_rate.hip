// Issue-rate microbenchmark for gfx950 (MI355X): cycles per wave64 instruction per SIMD for the instruction classes the
// raycast+graph kernel is made of, at 1 / 2 / 4 / 8 waves per SIMD (is the VALU stream full-rate?).
// Every block of 16 instructions uses 16 independent destination registers; a wave runs REPS blocks; the cost is
// Delta(s_memtime) of wave 0 / (instructions per wave x waves per SIMD).   Build + run: tools/micro/run_valu_rate.sh
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REPS 2048

#define X16(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)

// one asm statement = 16 instructions with 16 different destinations "+v"(r[i])
#define DEFINE_VOP(NAME, TEXT)                                                                        \
  __global__ void k_##NAME(unsigned long long* out, float seed) {                                     \
    float r[16];                                                                                      \
    for (int i = 0; i < 16; ++i) r[i] = seed + i + threadIdx.x;                                       \
    float a = seed * 1.0001f, b = seed * 0.9999f;                                                      \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                                             \
    for (int it = 0; it < REPS; ++it) {                                                               \
      asm volatile(TEXT(0) TEXT(1) TEXT(2) TEXT(3) TEXT(4) TEXT(5) TEXT(6) TEXT(7) TEXT(8) TEXT(9)    \
                   TEXT(10) TEXT(11) TEXT(12) TEXT(13) TEXT(14) TEXT(15)                              \
                   : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), \
                     "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15]) \
                   : "v"(a), "v"(b) : "vcc", "scc", "s40", "s41", "s42");                                              \
    }                                                                                                 \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                                             \
    float s = 0; for (int i = 0; i < 16; ++i) s += r[i];                                              \
    if (s == 12345.678f) out[1] = 1;                                                                  \
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) { atomicMin(&out[2], t0); atomicMax(&out[3], t1); }                                        \
  }

#define T_FMA(i)    "v_fma_f32 %" #i ", %16, %17, %" #i "\n"
#define T_MUL(i)    "v_mul_f32 %" #i ", %16, %" #i "\n"
#define T_ADD(i)    "v_add_f32 %" #i ", %16, %" #i "\n"
#define T_MIN(i)    "v_min_f32 %" #i ", %16, %" #i "\n"
#define T_MED3(i)   "v_med3_f32 %" #i ", %" #i ", %16, %17\n"
#define T_MOV(i)    "v_mov_b32 %" #i ", %16\n"
#define T_AND(i)    "v_and_b32 %" #i ", %16, %" #i "\n"
#define T_XOR(i)    "v_xor_b32 %" #i ", %16, %" #i "\n"
#define T_CNDV(i)   "v_cndmask_b32 %" #i ", %16, %" #i ", vcc\n"
#define T_CNDS(i)   "v_cndmask_b32 %" #i ", %16, %" #i ", s[40:41]\n"
#define T_CMPV(i)   "v_cmp_lt_f32 vcc, %16, %" #i "\n"
#define T_CMPS(i)   "v_cmp_lt_f32 s[40:41], %16, %" #i "\n"
#define T_CMPCND(i) "v_cmp_lt_f32 vcc, %16, %" #i "\n v_cndmask_b32 %" #i ", %17, %" #i ", vcc\n"
#define T_RCP(i)    "v_rcp_f32 %" #i ", %" #i "\n"
#define T_SQRT(i)   "v_sqrt_f32 %" #i ", %" #i "\n"
#define T_DSCALE(i) "v_div_scale_f32 %" #i ", vcc, %16, %17, %16\n"
#define T_DFMAS(i)  "v_div_fmas_f32 %" #i ", %" #i ", %16, %17\n"
#define T_DFIX(i)   "v_div_fixup_f32 %" #i ", %" #i ", %16, %17\n"
#define T_DPP(i)    "v_mov_b32_dpp %" #i ", %" #i " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define T_MINDPP(i) "v_min_f32_dpp %" #i ", %" #i ", %" #i " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define T_PKMUL(i)  "v_pk_mul_f32 %" #i ", %" #i ", %" #i "\n"
#define T_SAND(i)   "s_and_b64 s[40:41], s[40:41], vcc\n"
#define T_SNOP(i)   "s_nop 0\n"
#define T_VS(i)     "v_fma_f32 %" #i ", %16, %17, %" #i "\n s_and_b64 s[40:41], s[40:41], vcc\n"
#define T_BALLOT(i) "v_cmp_lt_f32 s[40:41], %16, %" #i "\n s_bcnt1_i32_b64 s42, s[40:41]\n"
#define T_BITOP(i)  "v_bfe_u32 %" #i ", %" #i ", 3, 5\n"
#define T_LSHL(i)   "v_lshlrev_b32 %" #i ", 2, %" #i "\n"
#define T_MAD24(i)  "v_mad_u32_u24 %" #i ", %16, %17, %" #i "\n"
#define T_ADDU(i)   "v_add_u32 %" #i ", %16, %" #i "\n"
#define T_READL(i)  "v_readlane_b32 s42, %" #i ", 3\n"

DEFINE_VOP(fma, T_FMA) DEFINE_VOP(mul, T_MUL) DEFINE_VOP(add, T_ADD) DEFINE_VOP(min, T_MIN) DEFINE_VOP(med3, T_MED3)
DEFINE_VOP(mov, T_MOV) DEFINE_VOP(and_, T_AND) DEFINE_VOP(xor_, T_XOR) DEFINE_VOP(cnd_vcc, T_CNDV) DEFINE_VOP(cnd_sgpr, T_CNDS)
DEFINE_VOP(cmp_vcc, T_CMPV) DEFINE_VOP(cmp_sgpr, T_CMPS) DEFINE_VOP(cmp_cnd_pair, T_CMPCND) DEFINE_VOP(rcp, T_RCP)
DEFINE_VOP(sqrt_, T_SQRT) DEFINE_VOP(div_scale, T_DSCALE) DEFINE_VOP(div_fmas, T_DFMAS) DEFINE_VOP(div_fixup, T_DFIX)
DEFINE_VOP(mov_dpp, T_DPP) DEFINE_VOP(min_dpp, T_MINDPP) DEFINE_VOP(s_and, T_SAND) DEFINE_VOP(s_nop0, T_SNOP)
DEFINE_VOP(fma_plus_salu, T_VS) DEFINE_VOP(cmp_bcnt_pair, T_BALLOT) DEFINE_VOP(bfe, T_BITOP) DEFINE_VOP(lshl, T_LSHL)
DEFINE_VOP(mad_u24, T_MAD24) DEFINE_VOP(add_u32, T_ADDU) DEFINE_VOP(readlane, T_READL)

// packed multiply needs 64-bit register pairs
__global__ void k_pk_mul(unsigned long long* out, float seed) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 r[16];
  for (int i = 0; i < 16; ++i) r[i] = f2{seed + i, seed - i};
  f2 a = f2{seed * 1.0001f, seed * 0.9999f};
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < REPS; ++it) {
#define PK(i) "v_pk_mul_f32 %" #i ", %16, %" #i "\n"
    asm volatile(PK(0) PK(1) PK(2) PK(3) PK(4) PK(5) PK(6) PK(7) PK(8) PK(9) PK(10) PK(11) PK(12) PK(13) PK(14) PK(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(a));
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0; for (int i = 0; i < 16; ++i) s += r[i].x + r[i].y;
  if (s == 12345.678f) out[1] = 1;
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) { atomicMin(&out[2], t0); atomicMax(&out[3], t1); }
}

// LDS: 16 independent ds_read_b128 (same address for all lanes = broadcast / distinct per lane), then one wait
__global__ void k_ds_read_b128(unsigned long long* out, float seed) {
  __shared__ float4 buf[1024];
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) buf[i] = make_float4(seed, seed, seed, seed);
  __syncthreads();
  float4 acc = make_float4(0, 0, 0, 0);
  const int lane = threadIdx.x & 63;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < REPS; ++it) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      float4 v = buf[(lane + j * 64 + it) & 1023];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[1] = 1;
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) { atomicMin(&out[2], t0); atomicMax(&out[3], t1); }
}
__global__ void k_ds_read_b128_bcast(unsigned long long* out, float seed) {
  __shared__ float4 buf[1024];
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) buf[i] = make_float4(seed, seed, seed, seed);
  __syncthreads();
  float4 acc = make_float4(0, 0, 0, 0);
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < REPS; ++it) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      float4 v = buf[(j * 7 + it) & 1023];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[1] = 1;
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) { atomicMin(&out[2], t0); atomicMax(&out[3], t1); }
}

struct Case { const char* name; void (*fn)(unsigned long long*, float); int per_block; };

int main() {
  fprintf(stderr, "start\n"); fflush(stderr);
  unsigned long long* d; hipMalloc(&d, 32);
  int ncu = 256; hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
  std::vector<Case> cases = {
#define C(n, k) {#n, k_##n, k}
    C(fma, 16), C(mul, 16), C(add, 16), C(pk_mul, 16), C(min, 16), C(med3, 16), C(mov, 16), C(and_, 16), C(xor_, 16), C(bfe, 16), C(lshl, 16),
    C(mad_u24, 16), C(add_u32, 16), C(cnd_vcc, 16), C(cnd_sgpr, 16), C(cmp_vcc, 16), C(cmp_sgpr, 16), C(cmp_cnd_pair, 32),
    C(rcp, 16), C(sqrt_, 16), C(div_scale, 16), C(div_fmas, 16), C(div_fixup, 16), C(mov_dpp, 16), C(min_dpp, 16), C(readlane, 16),
    C(s_and, 16), C(s_nop0, 16), C(fma_plus_salu, 32), C(cmp_bcnt_pair, 32), C(ds_read_b128, 16), C(ds_read_b128_bcast, 16)};
  struct Cfg { int block, grid_per_cu, wps; } cfgs[] = {{256, 1, 1}, {512, 1, 2}, {1024, 1, 4}, {1024, 2, 8}};
  printf("{\"device_cus\": %d, \"reps\": %d, \"unit\": \"shader cycles per wave64 instruction per SIMD\", \"rows\": [\n", ncu, REPS);
  bool first = true;
  fprintf(stderr, "runtime up, %d CUs\n", ncu); fflush(stderr);
  for (auto& c : cases) {
    double res[4];
    fprintf(stderr, "case %s\n", c.name); fflush(stderr);
    for (int k = 0; k < 4; ++k) {
      hipLaunchKernelGGL(c.fn, dim3(ncu * cfgs[k].grid_per_cu), dim3(cfgs[k].block), 0, 0, d, 1.5f);   // warm-up
      unsigned long long init[4] = {0, 0, ~0ull, 0};
      hipMemcpy(d, init, 32, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(c.fn, dim3(ncu * cfgs[k].grid_per_cu), dim3(cfgs[k].block), 0, 0, d, 1.5f);
      if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "kernel %s failed\n", c.name); return 1; }
      unsigned long long h[4]; hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
      // all waves of block 0 (first start -> last end); with two blocks per CU the co-resident block does the same work
      res[k] = (double)(h[3] - h[2]) / ((double)REPS * c.per_block * cfgs[k].wps);
    }
    printf("%s {\"op\": \"%s\", \"instr_per_block\": %d, \"w1\": %.2f, \"w2\": %.2f, \"w4\": %.2f, \"w8\": %.2f}", first ? " " : ",\n ", c.name,
           c.per_block, res[0], res[1], res[2], res[3]);
    first = false;
    fflush(stdout);
  }
  printf("\n]}\n");
  return 0;
}
