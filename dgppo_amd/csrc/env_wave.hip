// Wave-per-environment LiDAR env.step for the benchmark topologies (n_rays = 32, top_k = 8, compile-time n_agents / n_obs):
// the same outputs, bit for bit, as env_step_kernel / lidar_step_kernel in env_step.hip (and oracle/env_np.py), organised
// around INSTRUCTION COUNT — at 4096+ envs per launch the workgroup-per-env kernel is VALU/SALU-issue bound
// (profiles/README.md: 1 661 VALU + 891 SALU per wave, two waves per env).
//
// Reference arithmetic replaced: the same list as env_step.hip (LidarEnv.step dgppo/env/lidar_env/base.py:151-174 and
// everything it calls: env/utils.py:49-55,115-136, obstacle.py:62-105, lidar_spread.py:35-96, lidar_target.py:35-96,
// lidar_bicycle_target.py:92-118, lidar_env/base.py:180-271, utils/graph.py:35-44,212-247).
//
// What is different from lidar_step_kernel:
//   * one WAVE owns one environment at a time and loops over environments (persistent waves, 4 independent waves per
//     workgroup, each with its own LDS slab): no s_barrier anywhere — the phases of an env are ordered by the in-order
//     execution of a wave's DS instructions, so a compiler barrier between phases is all that is needed;
//   * every size is a template parameter: all loops unroll, index divisions fold, no dynamic-trip-count loop scaffolding;
//   * the constant parts of the node / state rows (indicator columns, zero padding, the pad row) are written into the
//     wave's LDS image ONCE; per env only agent / goal states and hit points are patched in, then the image is streamed out;
//   * segment tests are run on a COMPACTED candidate list: an (agent, obstacle) pair that is provably out of reach, and a ray
//     whose line provably misses the obstacle's bounding circle, can have no accepted segment (see "cull"); the surviving
//     (agent, obstacle, ray) triples — ~60 of 768 per env — are appended to a list in LDS (ballot + mbcnt) and processed 64
//     per pass, four segments each; results meet in a per-(agent, ray) LDS word through ds_min_u32 (alphas are >= +0, so
//     their bit patterns order like the floats);
//   * stable top-k straight from registers: ranks of the (many) missing rays by popcount of ballots, ranks of the (few)
//     hitting rays by a wave-uniform loop over the hitting lanes — no 32-key compare ladder, no keys in LDS;
//   * per-agent minima (cost, goal distances) by DPP row reductions instead of serial LDS loops.
//
// Cull (rigorous, keeps bit-exactness).  For agent position p and obstacle o let c_o be the mean of the four vertices the
// kernel actually uses and h_o the largest vertex distance from c_o.  If |p - c_o| > R + h_o + 0.05 (R = comm_radius =
// ray length) then every point of every edge P_m + beta e_m, beta in [0,1], is farther than R + 0.049 from p.  The test
// of obstacle.py:97-105 accepts a segment iff 0 <= na' <= |det| and 0 <= nb' <= |det| in fp32.  If additionally
// |det| >= 1e-4 for every (ray, edge) of that obstacle (checked per env from the ray table and the edge vectors:
// 0.5 |dir_r x e_m| >= 2e-4 for the two edge directions, the opposite edges differ by < 3e-7), then the accepted
// (alpha, beta) are within 4e-3 of the exact intersection parameters of the two lines (numerator errors < 3e-7), i.e.
// the exact intersection point would lie within 0.502 R of p and within 2e-3 of the edge — impossible at that distance.
// So no segment of a culled obstacle can be valid for either agent of the pair, det cannot be 0 or NaN there (NaN inputs
// fail the distance comparison and are never culled), and skipping the obstacle leaves alpha unchanged.  Obstacles whose
// edges are nearly parallel to a ray of the fan (about 1 in 70) are simply never culled.
// Ray-level cull, same argument: for such a "robust" obstacle an accepted (alpha, beta) has beta within 4e-3 of the exact
// parameter of the intersection X of the ray's LINE with the edge's line, so X lies within 4e-3 |e| < 2e-3 of the edge, i.e.
// within h_o + 2e-3 of c_o; and alpha within 4e-3 puts X no farther than 0.002 R behind p along the ray direction d.  Hence
// if the line misses the circle (c_o, h_o + 0.05) — |d x (c_o - p)| > h_o + 0.05 — or the circle lies wholly behind the
// ray origin — d . (c_o - p) < -(h_o + 0.05) — no segment of o is valid for this ray (both tests are evaluated in fp32
// with errors < 1e-6 against a slack of 0.048; a NaN fails both comparisons and keeps the triple).
#include "env_step.h"
#include <stdlib.h>

// minimum waves per SIMD the register allocator must leave room for (VGPR budget 512 / n, granule 8)
#ifndef DGPPO_WAVE_WPE
#define DGPPO_WAVE_WPE 5
#endif
#ifndef DGPPO_WAVE_WPE_COMPACT      // the kernels without the GraphsTuple image (training rollout, reset's sense pass)
#define DGPPO_WAVE_WPE_COMPACT 6
#endif
#ifndef DGPPO_WAVE_UNROLL_O
#define DGPPO_WAVE_UNROLL_O 8
#endif
#ifndef DGPPO_WAVE_UNROLL_P3
#define DGPPO_WAVE_UNROLL_P3 1
#endif
#define DGPPO_PRAGMA_(x) _Pragma(#x)
#define DGPPO_PRAGMA(x) DGPPO_PRAGMA_(x)

namespace {

// compiler-level ordering between phases of one wave: DS instructions of a wave execute in issue order, so a later
// ds_read observes an earlier ds_write of ANY lane of the same wave; only the compiler must not reorder them
#define WSYNC() asm volatile("" ::: "memory")
// phase markers: an assembler comment (free) and, in a -DDGPPO_STAMPS build (make stamps; tools/stamps_wave.py), an
// s_memtime stamp of wave 0's first environment
#ifdef DGPPO_STAMPS
__device__ unsigned long long g_wstamps[64];
__device__ unsigned long long g_wspan[3 * 8192];   // per wave: kernel entry, first env start, last env end (s_memtime)
#define PHASE(name, idx) do { asm volatile("; PHASE " name ::: "memory"); \
    if (blockIdx.x == 0 && threadIdx.x == 0 && stamp_on) g_wstamps[idx] = __builtin_amdgcn_s_memtime(); } while (0)
#elif defined(DGPPO_PHASE_MARKS)
// tools/isa_hist.py: assembler comments only (volatile keeps them ordered against the WSYNC fences; without a "memory"
// clobber they do not fence the scheduler themselves, so the instruction stream is the production one up to placement)
#define PHASE(name, idx) asm volatile("; PHASE " name)
#else
#define PHASE(name, idx) do { } while (0)    // (a marker with a "memory" clobber would also fence the scheduler)
#endif

constexpr int ceil4(int x) { return (x + 3) / 4 * 4; }

template <int SD, bool SPREAD, int NA, int NO>
struct WC {
  static constexpr int K = 8, R = 32, ND = SD + 3;
  static constexpr int NR = NA * R, NIT = NR / 64;
  static constexpr int NHIT = NA * K;
  static constexpr int N = 2 * NA + NHIT + 1, PAD = N - 1;
  static constexpr int GS = SPREAD ? NA : 1;
  static constexpr int E = NA * (NA + GS + K);
  static constexpr int NPAIR = NA * NO;
  static constexpr int NFAR = (NPAIR + 63) / 64;
  static_assert(NA % 2 == 0 && NA >= 2 && NA <= 16, "wave kernel: even n_agents <= 16 (two agents share a wave iteration)");
  static_assert((NA & (NA - 1)) == 0 && NA >= 4, "wave kernel: n_agents must be a power of two >= 4 (DPP group reductions)");
  static_assert(NO >= 1 && NO * 4 <= 64, "wave kernel: 1 <= n_obs <= 16");
  static_assert((NA * SD) % 4 == 0, "agent / goal rows are staged as float4");
  // the staged inputs agent | goal | obst | hits | act are contiguous float4 items: what is overlaid on them must not be larger
  static_assert(ceil4(3 * NA) <= NA * SD && 4 * NA + ceil4(NA * 2) <= NO * 16, "overlays exceed the staged input they reuse");
};

// per-wave LDS slab; every member is a multiple of 16 bytes so each starts 16-byte aligned
template <int SD, bool SPREAD, int NA, int NO, bool GRAPH>
struct alignas(16) WaveLds {
  using C = WC<SD, SPREAD, NA, NO>;
  float4 seg[NO * 4];                 // per segment: x3, y3, ex = x4 - x3, ey = y4 - y3
  float4 pc[NA * NO];                 // per (agent, obstacle): c_o - p, ray-cull threshold h_o + 0.05 (-1: pair out of reach, +inf: never cull)
  float4 circ[NO];                    // cull circle: cx, cy, (R + h + 0.05)^2, h + 0.05
  uint32_t alpha[NA * 32];            // per (agent, ray): bits of min alpha over the tested segments (1e6: no hit)
  uint32_t badm[ceil4(NA)];           // per agent: rays that met det == 0 / NaN (literal re-evaluation)
  // two scratch areas that are never live together share their bytes (LDS per wave decides how many waves a CU holds):
  union {
    uint16_t items[NA * NO * 32];     // P2: candidate (agent, obstacle, ray) triples: (agent * NO + obstacle) << 5 | ray
    struct {                          // P3:
      uint32_t tk[NA * 32];           //   per agent the keys of its hitting rays, compacted in ray order (0xFFFFFFFF: unused)
      uint32_t hcnt[ceil4(NA)];       //   per agent: number of hitting rays
      uint16_t hlist[NA * 32];        //   all hitting (agent, ray) of the env: agent << 10 | position in the agent's list << 5 | ray
    };
  };
  float next[ceil4(NA * SD)];         // state at t+1
  // inputs of the env, staged with ONE 16-byte load per lane: consecutive float4 items agent | goal | obst | hits | act
  union {                             // state at t; dead once the cost terms of graph_t are reduced, then:
    float agent[NA * SD];
    float red[ceil4(3 * NA)];         //   reward terms: d2g | indicator | ||a||^2
  };
  float goal[NA * SD];
  union {                             // obstacle records; dead once the segment / circle constants and the inside flags exist, then:
    float obst[NO * 16];
    struct {
      float sq[4 * NA];               //   squared minima awaiting ONE square root: agent-agent | agent-hit | goal-agent | ||a||^2
      float cost[ceil4(NA * 2)];
    };
  };
  float hits[NA * C::K * 2];          // hit points of graph_t until the cost terms are done, then those of graph_{t+1}
  float act[ceil4(NA * 2)];           // raw action (clipped where it is read)
  float fa[SD == 4 ? 4 : NA * 4];     // state2feat(next agent) / (goal): the identity for the double integrator, whose kernels
  float fg[SD == 4 ? 4 : NA * 4];     //   read `next` / `goal` instead (fa_ptr / fg_ptr)
  float ino[ceil4(NA * NO)];          // start-inside flags
  // [N, ND] image; the [N, SD] states are its leading columns (pad row: -1).  Only the kernels that emit the GraphsTuple
  // carry it: without it a wave's slab is a third smaller and more waves fit a CU
  float nodes[GRAPH ? ceil4(C::N * C::ND) : 4];
};

__device__ inline float raw_min(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ inline float raw_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// Rectangle.inside with r = 0 (obstacle.py:62-72 as called by env/utils.py:117): with r = 0 the corner-circle term
// sqrt(.) < 0 is never true (also not for NaN), so the test reduces to both local coordinates being inside
__device__ inline bool rect_inside_r0(const float* rec, float px, float py) {
  const float rel_x = px - rec[0], rel_y = py - rec[1];
  const float cs = rec[5], sn = rec[6];
  const float rel_xx = fabsf(rel_x * cs + rel_y * sn) - rec[2] / 2.0f;
  const float rel_yy = fabsf(rel_x * sn - rel_y * cs) - rec[3] / 2.0f;
  return (rel_xx < 0.0f) && (rel_yy < 0.0f);
}

template <int CTRL>
__device__ inline float dpp(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, false));
}
constexpr int DPP_XOR1 = 0xB1;      // quad_perm [1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;      // quad_perm [2,3,0,1]
constexpr int DPP_HALF_MIRROR = 0x141;
constexpr int DPP_ROW_MIRROR = 0x140;

// min over aligned groups of G lanes (G = 4, 8 or 16), jnp.min semantics (NaN propagates).  NaNs are practically
// never present, so the wave takes the 1-instruction-per-step path unless some lane holds one.
template <int G>
__device__ inline float group_min(float x) {
  if (__builtin_amdgcn_ballot_w64(x != x) == 0ull) {
    x = raw_min(x, dpp<DPP_XOR1>(x));
    x = raw_min(x, dpp<DPP_XOR2>(x));
    if constexpr (G >= 8) x = raw_min(x, dpp<DPP_HALF_MIRROR>(x));
    if constexpr (G >= 16) x = raw_min(x, dpp<DPP_ROW_MIRROR>(x));
  } else {
    x = nanmin(x, dpp<DPP_XOR1>(x));
    x = nanmin(x, dpp<DPP_XOR2>(x));
    if constexpr (G >= 8) x = nanmin(x, dpp<DPP_HALF_MIRROR>(x));
    if constexpr (G >= 16) x = nanmin(x, dpp<DPP_ROW_MIRROR>(x));
  }
  return x;
}

// MODE / GRAPH are compile-time: the training rollout (MODE_STEP, compact), the API step (MODE_STEP + GraphsTuple), the
// sense-only pass of reset and the materialise-only pass each get their own straight-line code
template <int SD, bool SPREAD, int NA, int NO, int WPB, int MODE, bool GRAPH>
__global__ void __launch_bounds__(WPB * 64) __attribute__((amdgpu_waves_per_eu(GRAPH ? DGPPO_WAVE_WPE : DGPPO_WAVE_WPE_COMPACT, 8))) lidar_wave_kernel(StepArgs a) {
  using C = WC<SD, SPREAD, NA, NO>;
  using LT = WaveLds<SD, SPREAD, NA, NO, GRAPH>;
  constexpr int K = C::K, ND = C::ND, N = C::N, PAD = C::PAD, NIT = C::NIT, GS = C::GS, E = C::E;
  extern __shared__ float4 smem4[];
  int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  LT& L = reinterpret_cast<LT*>(smem4)[wave];
#ifdef DGPPO_STAMPS
  const int gw = blockIdx.x * WPB + wave;
  if ((threadIdx.x & 63) == 0 && gw < 8192) g_wspan[3 * gw] = __builtin_amdgcn_s_memtime();
#endif
  const dgppo_env_cfg& c = a.cfg;
  constexpr bool do_dyn = (MODE == MODE_STEP);
  constexpr bool do_sense = (MODE != MODE_GRAPH);
  constexpr bool has_graph = GRAPH;
  const float sr = c.comm_radius;
  const int r = lane & 31;                      // this lane's ray, for the whole kernel
  const int hi_half = lane >> 5;                // which of the two agents of a wave iteration
  const float cr = do_sense ? a.ray_cos[r] : 0.0f, sn = do_sense ? a.ray_sin[r] : 0.0f;
  const uint32_t below = (1u << r) - 1u;        // lanes of my half with a smaller ray index

  // ---- input staging plan: item q (a float4) of the concatenation agent | goal | obst | hits | act of one env ----
  constexpr int I_AG = NA * SD / 4, I_OB = NO * 4, I_HI = (MODE != MODE_SENSE) ? NA * K * 2 / 4 : 0,
                I_AC = (MODE == MODE_STEP) ? NA * 2 / 4 : 0;
  constexpr int IT = 2 * I_AG + I_OB + I_HI + I_AC, NSLOT = (IT + 63) / 64;
  static_assert(MODE != MODE_STEP || (NA * 2) % 4 == 0, "the action block is staged as float4");
  const char* gsrc[NSLOT];
  uint32_t gstride[NSLOT];
  float4 pf[NSLOT];
#pragma unroll
  for (int j = 0; j < NSLOT; ++j) {
    const int q = j * 64 + lane;
    const char* base = reinterpret_cast<const char*>(a.agent);
    uint32_t stride = NA * SD * 4;
    int q0 = 0;
    if (q >= I_AG) { base = reinterpret_cast<const char*>(a.goal); q0 = I_AG; }
    if (q >= 2 * I_AG) { base = reinterpret_cast<const char*>(a.obst); stride = NO * 64; q0 = 2 * I_AG; }
    if (I_HI > 0 && q >= 2 * I_AG + I_OB) { base = reinterpret_cast<const char*>(a.hits); stride = NA * K * 8; q0 = 2 * I_AG + I_OB; }
    if (I_AC > 0 && q >= 2 * I_AG + I_OB + I_HI) { base = reinterpret_cast<const char*>(a.action); stride = NA * 8; q0 = 2 * I_AG + I_OB + I_HI; }
    gsrc[j] = base + (q - q0) * 16;
    gstride[j] = stride;
  }
  {
    const int b0 = blockIdx.x * WPB + wave;          // the first env's inputs fly while the images are initialised
    if (b0 < a.B) {
#pragma unroll
      for (int j = 0; j < NSLOT; ++j)
        if (j * 64 + lane < IT) pf[j] = *reinterpret_cast<const float4*>(gsrc[j] + (size_t)b0 * gstride[j]);
    }
  }
  // ---- once per wave: the constant part of the node / state images (lidar_env/base.py:236-264, graph.py:214-218) ----
  if (has_graph)
    for (int i = lane; i < ceil4(N * ND) / 4; i += 64) reinterpret_cast<float4*>(L.nodes)[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  WSYNC();
  if (has_graph)
    for (int node = lane; node < PAD; node += 64)
      L.nodes[node * ND + ((node < NA) ? SD + 2 : ((node < 2 * NA) ? SD + 1 : SD))] = 1.0f;
  WSYNC();

  for (int b = blockIdx.x * WPB + wave; b < a.B; b += gridDim.x * WPB) {
    // make the lane id opaque per environment: otherwise every lane-dependent LDS / global address of the (fully
    // unrolled) body is hoisted out of this loop and kept live — ~170 VGPRs, 3 waves per SIMD; recomputing them costs a
    // few dozen VALU instructions per env and keeps the kernel under 128
    asm volatile("" : "+v"(lane));
#ifdef DGPPO_STAMPS
    const bool stamp_on = (b == 0);
    if ((threadIdx.x & 63) == 0 && gw < 8192 && b == gw) g_wspan[3 * gw + 1] = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 0 && threadIdx.x == 0 && stamp_on) g_wstamps[40] = __builtin_amdgcn_s_memrealtime();   // 100 MHz
#endif
    PHASE("P0_stage", 0);
    // ---- P0: the env's inputs arrive in registers (issued one env ahead), one ds_write_b128 per lane and slot ----
#pragma unroll
    for (int j = 0; j < NSLOT; ++j)
      if (j * 64 + lane < IT) reinterpret_cast<float4*>(L.agent)[j * 64 + lane] = pf[j];
    {
      const int bn = b + gridDim.x * WPB;            // prefetch the next env of this wave while this one is processed
      if (bn < a.B) {
#pragma unroll
        for (int j = 0; j < NSLOT; ++j)
          if (j * 64 + lane < IT) pf[j] = *reinterpret_cast<const float4*>(gsrc[j] + (size_t)bn * gstride[j]);
      }
    }
    WSYNC();
    PHASE("P1a_dyn_seg_circ", 1);
    // ---- P1a: dynamics + features (lanes < NA), goal features (lanes 32..), segment constants, cull circles ----
    float asq = 0.0f;                              // lanes < NA: ||a_i||^2 (parked in a register: `sq` overlays the obstacle records)
    if (lane < NA) {
      const int i = lane;
      const float* x = L.agent + i * SD;
      float nx[SD];
      if (do_dyn) {
        const float u0 = clampf(L.act[i * 2], -1.0f, 1.0f), u1 = clampf(L.act[i * 2 + 1], -1.0f, 1.0f);   // env/base.py:84-86
        const float dt = c.dt, A = c.area_size;
        if constexpr (SD == 5) {  // lidar_bicycle_target.py:95-107
          const float theta = atan2f(x[3], x[2]);
          const float theta_next = theta + x[4] * u0 * dt * 10.0f;
          nx[0] = clampf(x[0] + x[4] * cosf(theta) * dt, 0.0f, A);
          nx[1] = clampf(x[1] + x[4] * sinf(theta) * dt, 0.0f, A);
          nx[2] = clampf(cosf(theta_next), -1.0f, 1.0f);
          nx[3] = clampf(sinf(theta_next), -1.0f, 1.0f);
          nx[SD - 1] = clampf(x[SD - 1] + u1 * dt * 10.0f, -0.5f, 0.5f);
        } else {                  // lidar_env/base.py:146-149
          const float vl = c.vel_limit;
          nx[0] = clampf(x[2] * dt + x[0], 0.0f, A);
          nx[1] = clampf(x[3] * dt + x[1], 0.0f, A);
          nx[2] = clampf((u0 * 10.0f) * dt + x[2], -vl, vl);
          nx[3] = clampf((u1 * 10.0f) * dt + x[3], -vl, vl);
        }
        asq = u0 * u0 + u1 * u1;                  // (||a||)^2 = fl(sqrt(.))^2: the root is taken below with the distances
      } else {
#pragma unroll
        for (int d = 0; d < SD; ++d) nx[d] = x[d];
      }
#pragma unroll
      for (int d = 0; d < SD; ++d) L.next[i * SD + d] = nx[d];
      if constexpr (SD != 4) state2feat<SD>(nx, L.fa + i * 4);
    }
    if constexpr (SD != 4)
      if (lane >= 32 && lane < 32 + NA) state2feat<SD>(L.goal + (lane - 32) * SD, L.fg + (lane - 32) * 4);
    const float* fa_ptr = (SD == 4) ? L.next : L.fa;
    const float* fg_ptr = (SD == 4) ? L.goal : L.fg;
    uint32_t robust_bits = 0;                      // bit o: every (ray, edge) determinant of obstacle o is >= 1e-4 in magnitude
    if (do_sense) {
      if (lane < NO * 4) {
        const int o = lane >> 2, m = lane & 3, mm = (m + 3) & 3;
        const float* P = L.obst + o * 16 + 8;
        L.seg[lane] = make_float4(P[2 * m], P[2 * m + 1], P[2 * mm] - P[2 * m], P[2 * mm + 1] - P[2 * m + 1]);
      }
      if (lane < NO) {
        const float* P = L.obst + lane * 16 + 8;
        const float cx = ((P[0] + P[2]) + (P[4] + P[6])) * 0.25f, cy = ((P[1] + P[3]) + (P[5] + P[7])) * 0.25f;
        float h2 = 0.0f;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const float dx = P[2 * m] - cx, dy = P[2 * m + 1] - cy;
          h2 = fmaxf(h2, dx * dx + dy * dy);
        }
        const float hm = sqrtf(h2) + 0.05f, thr = sr + hm;
        L.circ[lane] = make_float4(cx, cy, thr * thr, hm);
      }
#pragma unroll
      for (int o = 0; o < NO; ++o) {               // lanes: edge direction m = lane >> 5 (0, 1), ray = lane & 31
        const int m = hi_half, mm = (m + 3) & 3;
        const float* P = L.obst + o * 16 + 8;
        const float ex = P[2 * mm] - P[2 * m], ey = P[2 * mm + 1] - P[2 * m + 1];
        const float cc = cr * ey - sn * ex;
        const bool weak = !(fabsf(cc) >= 4e-4f);   // also true for NaN
        if (__builtin_amdgcn_ballot_w64(weak) == 0ull) robust_bits |= (1u << o);
      }
    }
    WSYNC();
    PHASE("P1b_far_as", 2);
    // ---- P1b: (agent, obstacle) start-inside flags + cull bits, (agent, segment) terms, pre-step distances ----
    uint64_t far[C::NFAR];
    if (do_sense) {
#pragma unroll
      for (int w = 0; w < C::NFAR; ++w) {
        const int q = w * 64 + lane;
        bool isfar = false;
        if (q < C::NPAIR) {
          const int i = q / NO, o = q - i * NO;
          const float px = L.next[i * SD], py = L.next[i * SD + 1];
          L.ino[q] = rect_inside_r0(L.obst + o * 16, px, py) ? 1.0f : 0.0f;
          const float4 cc = L.circ[o];
          const float dx = px - cc.x, dy = py - cc.y;
          const bool robust = (robust_bits >> o) & 1u;
          isfar = (dx * dx + dy * dy > cc.z) && robust;
          // ray-level cull threshold of this pair: out of reach -> every ray is culled; non-robust obstacle -> none is
          L.pc[q] = make_float4(cc.x - px, cc.y - py, isfar ? -1.0f : (robust ? cc.w : __builtin_inff()), 0.0f);
        }
        far[w] = __builtin_amdgcn_ballot_w64(isfar);
      }
    } else {
#pragma unroll
      for (int w = 0; w < C::NFAR; ++w) far[w] = 0ull;
    }
    PHASE("P1c_cost_terms", 3);
    if (do_dyn) {
      WSYNC();                                       // `sq` overlays the obstacle records: every read of them is issued by now
      if (lane < NA) L.sq[3 * NA + lane] = asq;
    }
    if (do_dyn) {  // cost and reward terms on the PRE-step graph (lidar_env/base.py:170-171,180-207; lidar_spread.py:35-52)
      // The reference takes min_j sqrt(s_j); the correctly rounded square root is monotone, so min_j fl(sqrt(s_j)) ==
      // fl(sqrt(min_j s_j)) bit for bit: reduce the SQUARED distances and take one root per agent afterwards (the root is
      // ~14 instructions per wave whatever the number of active lanes).
#pragma unroll
      for (int q0 = 0; q0 < NA * NA; q0 += 64) {    // agent-agent: min over j != i of ||p_i - p_j||^2 (the diagonal's
        const int q = q0 + lane, i = q / NA, j = q - i * NA;   // sqrt(0) + 1e6 enters after the root)
        float d = __builtin_inff();
        if (q < NA * NA && j != i) {
          const float dx = L.agent[i * SD] - L.agent[j * SD], dy = L.agent[i * SD + 1] - L.agent[j * SD + 1];
          d = dx * dx + dy * dy;
        }
        const float md2 = group_min<NA>(d);
        if (q < NA * NA && j == 0) L.sq[i] = md2;
      }
#pragma unroll
      for (int q0 = 0; q0 < NA * K; q0 += 64) {     // agent-hit: min over the agent's k hit points of graph_t
        const int q = q0 + lane, i = q / K;
        float d = __builtin_inff();
        if (q < NA * K) {
          const float dx = L.hits[q * 2] - L.agent[i * SD], dy = L.hits[q * 2 + 1] - L.agent[i * SD + 1];
          d = dx * dx + dy * dy;
        }
        const float mo2 = group_min<K>(d);
        if (q < NA * K && (q & (K - 1)) == 0) L.sq[NA + i] = mo2;
      }
      if constexpr (SPREAD) {                        // each goal: distance to the nearest agent
#pragma unroll
        for (int q0 = 0; q0 < NA * NA; q0 += 64) {
          const int q = q0 + lane, g = q / NA, j = q - g * NA;
          float d = __builtin_inff();
          if (q < NA * NA) {
            const float dx = L.goal[g * SD] - L.agent[j * SD], dy = L.goal[g * SD + 1] - L.agent[j * SD + 1];
            d = dx * dx + dy * dy;
          }
          const float g2 = group_min<NA>(d);
          if (q < NA * NA && j == 0) L.sq[2 * NA + g] = g2;
        }
      } else {                                       // paired goal
        if (lane < NA) {
          const int g = lane;
          const float dx = L.goal[g * SD] - L.agent[g * SD], dy = L.goal[g * SD + 1] - L.agent[g * SD + 1];
          L.sq[2 * NA + g] = dx * dx + dy * dy;
        }
      }
      WSYNC();
      static_assert(4 * NA <= 64, "one lane per pending square root");
      if (lane < 4 * NA) {
        const float rt = sqrtf(L.sq[lane]);          // the only square root of this phase
        const int kind = lane / NA, i = lane - kind * NA;
        if (kind == 0) {        // agent_cost = 2r - min(min_{j != i} dist, sqrt(0) + 1e6)   [jnp.min: NaN propagates]
          const float md = nanmin(rt, 1e6f);
          const float agent_cost = c.two_car_radius - md;
          const float c0 = (agent_cost <= 0.0f) ? agent_cost - 0.5f : agent_cost + 0.5f;
          L.cost[i * 2] = clampf_nan(c0, -1.0f, 1.0f);
        } else if (kind == 1) {
          const float obs_cost = c.car_radius - rt;
          const float c1 = (obs_cost <= 0.0f) ? obs_cost - 0.5f : obs_cost + 0.5f;
          L.cost[i * 2 + 1] = clampf_nan(c1, -1.0f, 1.0f);
        } else if (kind == 2) {
          L.red[i] = rt;
          L.red[NA + i] = (rt > c.dist2goal) ? 1.0f : 0.0f;
        } else {
          L.red[2 * NA + i] = rt * rt;
        }
      }
    }
    WSYNC();
    PHASE("P1d_reward", 4);
    if (do_dyn) {
      // reward: three sequential sums in index order (the oracle's seq_sum), one per lane, then combined on lane 0
      float s = 0.0f;
      if (lane < 3) {
        s = L.red[lane * NA];
#pragma unroll
        for (int g = 1; g < NA; ++g) s = s + L.red[lane * NA + g];
      }
      const float s1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s), 0));
      const float s2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s), 1));
      const float s3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s), 2));
      if (lane == 0) {
        float rw = 0.0f;
        rw = rw - (s1 / (float)NA) * 0.01f;
        rw = rw - (s2 / (float)NA) * 0.001f;
        rw = rw - (s3 / (float)NA) * 0.0001f;
        a.reward[b] = rw;
      }
      if (lane < NA * 2) a.cost[(size_t)b * NA * 2 + lane] = L.cost[lane];
    }
    // ---- early outputs: everything that does not depend on the ray-cast leaves now, so that these stores drain while
    //      the segment tests run (at one env per wave all waves reach their stores at the same time otherwise) ----
    if (a.next_agent != nullptr) {
      float4* o4 = reinterpret_cast<float4*>(a.next_agent + (size_t)b * NA * SD);
      for (int i = lane; i < NA * SD / 4; i += 64) o4[i] = reinterpret_cast<const float4*>(L.next)[i];
    }
    if (has_graph) {                     // agent / goal rows of the images
      for (int q = lane; q < NA * SD; q += 64) {
        const int node = q / SD, d = q - node * SD;
        const float va = L.next[q], vg = L.goal[q];
        L.nodes[node * ND + d] = va;
        L.nodes[(NA + node) * ND + d] = vg;
      }
    }
    if (has_graph) {
      float4* edges = reinterpret_cast<float4*>(a.g.edges) + (size_t)b * E;
      int32_t* recv = a.g.receivers + (size_t)b * E;
      int32_t* send = a.g.senders + (size_t)b * E;
#pragma unroll
      for (int q0 = 0; q0 < NA * NA; q0 += 64) {     // agent-agent block
        const int q = q0 + lane;
        if (q < NA * NA) {
          const int i = q / NA, j = q - i * NA;
          const float4 fi = reinterpret_cast<const float4*>(fa_ptr)[i], fj = reinterpret_cast<const float4*>(fa_ptr)[j];
          const float dx = L.next[i * SD] - L.next[j * SD], dy = L.next[i * SD + 1] - L.next[j * SD + 1];
          // (sqrt(s) + (i == j ? comm_radius + 1 : 0)) < comm_radius  <=>  i != j && s < T, T = the smallest float whose
          // correctly rounded root reaches comm_radius (host-computed, sqrt_threshold()); false for NaN on both sides
          const bool mask = (i != j) && (dx * dx + dy * dy < a.thr2_comm);
          edges[q] = make_float4(fi.x - fj.x, fi.y - fj.y, fi.z - fj.z, fi.w - fj.w);
          recv[q] = mask ? i : PAD;
          send[q] = mask ? j : PAD;
        }
      }
#pragma unroll
      for (int q0 = 0; q0 < NA * GS; q0 += 64) {     // agent-goal block (all-true mask)
        const int q = q0 + lane;
        if (q < NA * GS) {
          const int i = SPREAD ? q / NA : q, g = SPREAD ? q - i * NA : q;
          const float4 fi = reinterpret_cast<const float4*>(fa_ptr)[i], fg = reinterpret_cast<const float4*>(fg_ptr)[g];
          edges[NA * NA + q] = make_float4(fi.x - fg.x, fi.y - fg.y, fi.z - fg.z, fi.w - fg.w);
          recv[NA * NA + q] = i;
          send[NA * NA + q] = NA + g;
        }
      }
      int32_t* nty = a.g.node_type + (size_t)b * N;
#pragma unroll
      for (int q0 = 0; q0 < N; q0 += 64) {
        const int node = q0 + lane;
        if (node < N) nty[node] = (node < NA) ? 0 : ((node < 2 * NA) ? 1 : ((node < PAD) ? 2 : -1));
      }
      if (lane == 0) { a.g.n_node[b] = N; a.g.n_edge[b] = E; }
    }
    PHASE("P2_rays", 5);
    // ---- P2: segment tests (obstacle.py:97-105) on the compacted candidate list -> alpha[agent][ray] ----
    if (do_sense) {
      // P2a: reset the per-(agent, ray) minima; candidate triples by ballot + prefix count, two agents x 32 rays per step
      for (int q = lane; q < NA * 32 / 4; q += 64)
        reinterpret_cast<uint4*>(L.alpha)[q] = make_uint4(MISS_BITS, MISS_BITS, MISS_BITS, MISS_BITS);
      if (lane < ceil4(NA)) L.badm[lane] = 0u;
      // Every lane first collects, in a bit mask, which of its (step, obstacle) triples survive the two culls — straight-line
      // arithmetic with independent LDS reads, no ballot -> scalar -> vector round trip per obstacle — then ONE wave-wide
      // prefix sum of the per-lane counts places the survivors in the list (their order is irrelevant: results meet in a min).
      int total = 0;                                   // wave-uniform
      constexpr int GSTEPS = (2 * NO * NIT <= 32) ? NIT : ((32 / (2 * NO)) >= 1 ? (32 / (2 * NO)) : 1);   // steps per 32-bit mask
      static_assert(2 * NO <= 32, "one step's triples must fit the lane mask");
#pragma unroll
      for (int g0 = 0; g0 < NIT; g0 += GSTEPS) {
        uint32_t nmask = 0u;                           // bit (it - g0) * 2 NO + o: (agent (2 it + half), obstacle o, my ray) survives
#pragma unroll
        for (int it = g0; it < g0 + GSTEPS && it < NIT; ++it) {
#pragma unroll
          for (int o = 0; o < NO; ++o) {
            const int qa = (it * 2) * NO + o, qb = (it * 2 + 1) * NO + o;
            const bool fa_ = (far[qa >> 6] >> (qa & 63)) & 1ull, fb_ = (far[qb >> 6] >> (qb & 63)) & 1ull;
            if (fa_ && fb_) continue;                  // wave-uniform: neither agent of this step can reach obstacle o
            const float4 pc = L.pc[(it * 2 + hi_half) * NO + o];
            const float cross = cr * pc.y - sn * pc.x, dot = cr * pc.x + sn * pc.y;
            const bool need = !(fabsf(cross) > pc.z) && !(dot < -pc.z);        // NaN keeps the triple
            nmask |= need ? (1u << ((it - g0) * 2 * NO + o)) : 0u;
          }
        }
        // inclusive prefix sum of the counts over the 64 lanes: 4 shifts inside each row of 16, then the row totals
        const int cnt = __popc(nmask);
        int inc = cnt;
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x111, 0xF, 0xF, false);     // row_shr:1
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x112, 0xF, 0xF, false);     // row_shr:2
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x114, 0xF, 0xF, false);     // row_shr:4
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x118, 0xF, 0xF, false);     // row_shr:8
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x142, 0xA, 0xF, false);     // row_bcast:15 -> rows 1, 3
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x143, 0xC, 0xF, false);     // row_bcast:31 -> rows 2, 3
        int pos = total + inc - cnt;
        total += __builtin_amdgcn_readlane(inc, 63);
        // code of bit b: pair index (2 (g0 + b / 2NO) + half) NO + b % 2NO  ==  2 g0 NO + half NO + b   (b % 2NO < NO by construction)
        const uint32_t base_code = ((uint32_t)((2 * g0 + hi_half) * NO) << 5) | (uint32_t)r;
        while (__builtin_amdgcn_ballot_w64(nmask != 0u) != 0ull) {              // trips: the largest per-lane count (<= 2 NO GSTEPS)
          if (nmask != 0u) {
            const int bit = __builtin_ctz(nmask);
            L.items[pos] = (uint16_t)(base_code + ((uint32_t)bit << 5));
            pos += 1;
            nmask &= nmask - 1u;
          }
        }
      }
      WSYNC();
      PHASE("P2b_dense", 6);
      // P2b: 64 candidates per pass, four segments each.  min over segments and obstacles through ds_min_u32 on the bit
      // patterns (every accepted alpha is >= +0); a lane that meets det == 0 / NaN only marks its (agent, ray).
#pragma unroll 1
      for (int k0 = 0; k0 < total; k0 += 64) {
        const int k = k0 + lane;
        // all 64 lanes fetch an item (lanes past the end re-read item 0): the ray's direction comes from the lane that owns
        // the ray (lane rr holds cos / sin of ray rr) through ds_bpermute, which needs its source lanes active
        const uint32_t item = L.items[k < total ? k : 0];
        const int rr = (int)(item & 31u);
        const float crr = __int_as_float(__builtin_amdgcn_ds_bpermute(rr << 2, __float_as_int(cr)));
        const float snr = __int_as_float(__builtin_amdgcn_ds_bpermute(rr << 2, __float_as_int(sn)));
        if (k < total) {
          const int pair = (int)(item >> 5);
          const int i = pair / NO, o = pair - i * NO;
          const float x1 = L.next[i * SD], y1 = L.next[i * SD + 1];
          const float x2 = x1 + crr * sr, y2 = y1 + snr * sr;
          const float dx12 = x1 - x2, dy12 = y1 - y2, ndy12 = -dy12;
          float naf[4], adet[4];
          bool valid[4];
          float dprod = 1.0f;
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            const float4 sg = L.seg[o * 4 + m];
            const float ax = x1 - sg.x, ay = y1 - sg.y;
            const float na = sg.w * ax - sg.z * ay;
            const float det0 = dx12 * sg.w - dy12 * sg.z;
            const float nb = ndy12 * ax + dx12 * ay;
            // flip both numerators by the sign of det: (na/det, nb/det) == (na'/|det|, nb'/|det|), exactly
            const uint32_t sb = __float_as_uint(det0) & 0x80000000u;
            naf[m] = __uint_as_float(__float_as_uint(na) ^ sb);
            const float nbf = __uint_as_float(__float_as_uint(nb) ^ sb);
            adet[m] = __builtin_amdgcn_fmed3f(fabsf(det0), 1e-7f, 1e7f);      // clip(|det|, 1e-7, 1e7)
            const float mn = raw_min(naf[m], nbf), mx = raw_max(naf[m], nbf);
            // 0 <= q <= 1 for both quotients  <=>  0 <= min(na', nb') and max(na', nb') <= |det|   (-0 >= 0 holds, like -0/d >= 0)
            valid[m] = (mn >= 0.0f) && (mx <= adet[m]);
            dprod = dprod * det0;
          }
          // det == 0 or NaN on any of the four segments (a product that underflows to 0 only sends the ray to the literal
          // path below, which is always right)
          const bool bad = !(dprod != 0.0f);
          float amin = 1e6f;
          // one branch for the pass: when some lane hits some segment, the four correctly rounded divisions are issued
          // together (independent chains interleave)
          if (__builtin_amdgcn_ballot_w64(valid[0] || valid[1] || valid[2] || valid[3]) != 0ull) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
              const float qa_ = naf[m] / adet[m] + 0.0f;                      // v*alpha + (1-v)*1e6 with v = 1 (turns -0 into +0)
              const float al = valid[m] ? qa_ : 1e6f;
              amin = raw_min(amin, al);
            }
          }
          if (amin < 1e6f) atomicMin(&L.alpha[i * 32 + rr], __float_as_uint(amin));
          if (bad) atomicOr(&L.badm[i], 1u << rr);
        }
      }
      WSYNC();
      PHASE("P2_slowcheck", 10);
      // rays that met det == 0 / NaN: literal reference arithmetic over every segment (see lidar_step_kernel); practically never
      if (__builtin_amdgcn_ballot_w64(lane < NA && L.badm[lane < NA ? lane : 0] != 0u) != 0ull) {
#pragma unroll 1
        for (int it = 0; it < NIT; ++it) {
          const int i = it * 2 + hi_half;
          if ((L.badm[i] >> r) & 1u) {
            const float x1 = L.next[i * SD], y1 = L.next[i * SD + 1];
            const float x2 = x1 + cr * sr, y2 = y1 + sn * sr;
            const float dx12 = x1 - x2, dy12 = y1 - y2, ndy12 = -dy12;
            float lmin = 1e6f;
            bool any_nan = false;
#pragma unroll 1
            for (int q = 0; q < NO * 4; ++q) {
              const float4 sgq = L.seg[q];
              const float ax = x1 - sgq.x, ay = y1 - sgq.y;
              const float na = sgq.w * ax - sgq.z * ay;
              const float det0 = dx12 * sgq.w - dy12 * sgq.z;
              const float nb = ndy12 * ax + dx12 * ay;
              const float sgn = (det0 > 0.0f) ? 1.0f : ((det0 < 0.0f) ? -1.0f : det0);
              const float dz = sgn * fminf(fmaxf(fabsf(det0), 1e-7f), 1e7f);
              const float aq = na / dz, bq = nb / dz;
              const float v = ((aq <= 1.0f) && (aq >= 0.0f) && (bq <= 1.0f) && (bq >= 0.0f)) ? 1.0f : 0.0f;
              const float al = v * aq + (1.0f - v) * 1e6f;
              any_nan = any_nan || (al != al);
              lmin = fminf(lmin, al);
            }
            L.alpha[i * 32 + r] = __float_as_uint(any_nan ? __builtin_nanf("") : lmin);
          }
        }
        WSYNC();
      }
      PHASE("P3_begin", 11);
      // ---- P3: per step, two agents x 32 rays: stable top-k (env/utils.py:132-136) -> hit points patched into hnext and
      //      the node / state images ----
      // Common case (no NaN alpha in the env): misses and inside-an-obstacle agents know their rank from a prefix count and
      // write their point at once; the (few) hitting rays of ALL agents go to one list and are ranked 64 per pass against
      // their agent's compacted key list, ties broken by ray order.
      bool nan_any = false;
      int n_hits = 0, cmaxw = 0;                        // wave-uniform: hitting rays of the env, largest per-agent count
      for (int q = lane; q < NA * 32 / 4; q += 64)
        reinterpret_cast<uint4*>(L.tk)[q] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
      WSYNC();                                         // the keys below land after this fill (DS operations of a wave are in order)
DGPPO_PRAGMA(unroll DGPPO_WAVE_UNROLL_P3)
      for (int it = 0; it < NIT; ++it) {
        const int i = it * 2 + hi_half;
        const float x1 = L.next[i * SD], y1 = L.next[i * SD + 1];
        const float x2 = x1 + cr * sr, y2 = y1 + sn * sr;
        float is_in = 0.0f;
#pragma unroll
        for (int o = 0; o < NO; ++o) is_in = fmaxf(is_in, L.ino[i * NO + o]);
        float ar = __uint_as_float(L.alpha[i * 32 + r]);
        ar = ar * (1.0f - is_in);
        nan_any = nan_any || (ar != ar);
        const uint32_t kr = __float_as_uint(ar);
        // an agent that starts inside an obstacle has all 32 alphas == +0 (alpha * (1 - is_in)): ranks are the ray indices
        const bool allz = is_in != 0.0f;
        const bool hit = (kr < MISS_BITS) && !allz;
        const uint64_t Lm = __builtin_amdgcn_ballot_w64(hit);
        const uint32_t lo = (uint32_t)Lm, hi = (uint32_t)(Lm >> 32);
        const int n_lo = __popc(lo), n_hi = __popc(hi);
        const int pfx = (int)__builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u));   // hits of the step below my lane
        const int p = pfx - (hi_half ? n_lo : 0);                                                  // ... of my agent
        if (lane == 0) { L.hcnt[it * 2] = (uint32_t)n_lo; L.hcnt[it * 2 + 1] = (uint32_t)n_hi; }
        // misses keep ray order behind all hits: rank = #hits + #misses below = #hits + (r - p)
        const int rank_direct = allz ? r : ((hi_half ? n_hi : n_lo) + r - p);
        if (!hit && rank_direct < K) {
          const float hx = x1 + (x2 - x1) * ar, hy = y1 + (y2 - y1) * ar;
          const int hq = i * K + rank_direct, node = 2 * NA + hq;
          L.hits[hq * 2] = hx; L.hits[hq * 2 + 1] = hy;
          if (has_graph) { L.nodes[node * ND] = hx; L.nodes[node * ND + 1] = hy; }
        }
        if (hit) {
          L.tk[i * 32 + p] = kr;
          L.hlist[n_hits + pfx] = (uint16_t)((i << 10) | (p << 5) | r);
        }
        n_hits += n_lo + n_hi;
        cmaxw = cmaxw > n_lo ? cmaxw : n_lo;
        cmaxw = cmaxw > n_hi ? cmaxw : n_hi;
      }
      WSYNC();
      if (__builtin_amdgcn_ballot_w64(nan_any) == 0ull) {
#pragma unroll 1
        for (int k0 = 0; k0 < n_hits; k0 += 64) {
          const int k = k0 + lane;
          const uint32_t rec = L.hlist[k < n_hits ? k : 0];
          const int rr = (int)(rec & 31u);
          const float crr = __int_as_float(__builtin_amdgcn_ds_bpermute(rr << 2, __float_as_int(cr)));
          const float snr = __int_as_float(__builtin_amdgcn_ds_bpermute(rr << 2, __float_as_int(sn)));
          if (k < n_hits) {
            const int i = (int)(rec >> 10), p = (int)((rec >> 5) & 31u);
            const uint32_t kr = L.tk[i * 32 + p];
            int rank = 0;
#pragma unroll 1
            for (int cb = 0; cb < cmaxw; cb += 4) {     // wave-uniform bound; slots past an agent's count hold 0xFFFFFFFF
              const uint4 ks = *reinterpret_cast<const uint4*>(&L.tk[i * 32 + cb]);
              rank += ((ks.x < kr) || (ks.x == kr && cb + 0 < p)) ? 1 : 0;
              rank += ((ks.y < kr) || (ks.y == kr && cb + 1 < p)) ? 1 : 0;
              rank += ((ks.z < kr) || (ks.z == kr && cb + 2 < p)) ? 1 : 0;
              rank += ((ks.w < kr) || (ks.w == kr && cb + 3 < p)) ? 1 : 0;
            }
            if (rank < K) {
              const float x1 = L.next[i * SD], y1 = L.next[i * SD + 1];
              const float x2 = x1 + crr * sr, y2 = y1 + snr * sr;
              const float ar = __uint_as_float(kr);
              const float hx = x1 + (x2 - x1) * ar, hy = y1 + (y2 - y1) * ar;
              const int hq = i * K + rank, node = 2 * NA + hq;
              L.hits[hq * 2] = hx; L.hits[hq * 2 + 1] = hy;
              if (has_graph) { L.nodes[node * ND] = hx; L.nodes[node * ND + 1] = hy; }
            }
          }
        }
      } else {
        // ---- a NaN alpha somewhere (ray parallel to an edge): three classes (hit, miss, NaN), one step
        //      (two agents) at a time; rewrites every slot of every agent ----
        WSYNC();
  #pragma unroll 1
        for (int it = 0; it < NIT; ++it) {
          const int i = it * 2 + hi_half;
          const float x1 = L.next[i * SD], y1 = L.next[i * SD + 1];
          const float x2 = x1 + cr * sr, y2 = y1 + sn * sr;
          float is_in = 0.0f;
  #pragma unroll
          for (int o = 0; o < NO; ++o) is_in = fmaxf(is_in, L.ino[i * NO + o]);
          float ar = __uint_as_float(L.alpha[i * 32 + r]);
          ar = ar * (1.0f - is_in);
          PHASE("P3_topk", 12 + it * 3 + 1);
          int rank;
          if (__builtin_amdgcn_ballot_w64(ar != ar) == 0ull) {
            // ---- common case, no NaN among the 64 alphas: two classes, hit (key < 1e6) and miss (key == 1e6) ----
            const uint32_t kr = __float_as_uint(ar);
            const bool hit = kr < MISS_BITS;
            const uint64_t Lm = __builtin_amdgcn_ballot_w64(hit);
            const uint32_t lo = (uint32_t)Lm, hi = (uint32_t)(Lm >> 32);
            const int n_lo = __popc(lo), n_hi = __popc(hi);
            // hits of my half with a smaller ray index (prefix count over the ballot, minus the other half's total)
            const int p = (int)__builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u)) - (hi_half ? n_lo : 0);
            // an agent that starts inside an obstacle has all 32 alphas == +0 (alpha * (1 - is_in)): ranks are the ray indices
            const bool allz = is_in != 0.0f;
            const bool z_lo = __builtin_amdgcn_readlane((int)allz, 0) != 0, z_hi = __builtin_amdgcn_readlane((int)allz, 32) != 0;
            const int c_lo = z_lo ? 0 : n_lo, c_hi = z_hi ? 0 : n_hi;
            const int cmax = c_lo > c_hi ? c_lo : c_hi;
            // misses keep ray order behind all hits: rank = #hits + #misses below = #hits + (r - p)
            rank = allz ? r : ((hi_half ? n_hi : n_lo) + r - p);
            if (cmax > 0) {                            // wave-uniform: some agent of this step has hits to order
              const bool isL = hit && !allz;
              // every hitting lane puts its key into its half's slot list, compacted in ray order; each lane then counts the
              // slots with a smaller key, four slots per trip of a wave-uniform loop.  Unused slots hold 0xFFFFFFFF.
              L.tk[lane] = 0xFFFFFFFFu;
              WSYNC();
              if (isL) L.tk[hi_half * 32 + p] = kr;
              WSYNC();
              int rlt = 0;
  #pragma unroll 1
              for (int cb = 0; cb < cmax; cb += 4) {
                const uint4 ks = *reinterpret_cast<const uint4*>(&L.tk[hi_half * 32 + cb]);
                rlt += (ks.x < kr ? 1 : 0) + (ks.y < kr ? 1 : 0) + (ks.z < kr ? 1 : 0) + (ks.w < kr ? 1 : 0);
              }
              // bit-identical alphas among the hits need the index tie-break: they show up as two lanes claiming the same
              // strict rank (claim slot rlt with p, read it back); practically never taken
              WSYNC();
              if (isL) L.tk[hi_half * 32 + rlt] = (uint32_t)p;
              WSYNC();
              const bool lost = isL && (L.tk[hi_half * 32 + rlt] != (uint32_t)p);
              if (__builtin_amdgcn_ballot_w64(lost) != 0ull) {
                WSYNC();
                L.tk[lane] = 0xFFFFFFFFu;
                WSYNC();
                if (isL) L.tk[hi_half * 32 + p] = kr;
                WSYNC();
                rlt = 0;
  #pragma unroll 1
                for (int sl = 0; sl < cmax; ++sl) {
                  const uint32_t kj = L.tk[hi_half * 32 + sl];
                  rlt += (kj < kr || (kj == kr && sl < p)) ? 1 : 0;
                }
                WSYNC();
              }
              if (isL) rank = rlt;
            }
          } else {
            // ---- a NaN alpha somewhere (ray parallel to an edge): three classes, general code ----
            // sort key: float bits (alphas are >= +0), NaN -> max.  Classes: L (hit, key < 1e6), M (miss, key == 1e6), H (NaN)
            const uint32_t kr = (ar != ar) ? 0xFFFFFFFFu : __float_as_uint(ar);
            const uint64_t Lm = __builtin_amdgcn_ballot_w64(kr < MISS_BITS);
            const uint64_t Mm = __builtin_amdgcn_ballot_w64(kr == MISS_BITS);
            const uint64_t Zm = __builtin_amdgcn_ballot_w64(kr == 0u);
            const uint32_t lo = (uint32_t)Lm, hi = (uint32_t)(Lm >> 32);
            const uint32_t myL = hi_half ? hi : lo;
            const uint32_t myM = hi_half ? (uint32_t)(Mm >> 32) : (uint32_t)Mm;
            const int cntL = __popc(myL);
            // misses keep ray order behind all hits; NaNs behind the misses (stable ascending sort)
            const int rank_other = (kr == MISS_BITS) ? cntL + __popc(myM & below)
                                                     : cntL + __popc(myM) + __popc(~(myL | myM) & below);
            // an agent that starts inside an obstacle has all 32 alphas == 0: ranks are the ray indices
            const bool z_lo = ((uint32_t)Zm == 0xFFFFFFFFu), z_hi = ((uint32_t)(Zm >> 32) == 0xFFFFFFFFu);
            const bool allz = hi_half ? z_hi : z_lo;
            const bool isL = (kr < MISS_BITS) && !allz;
            // Ranks of the hitting rays: every hitting lane puts its key into its half's slot list, compacted in ray order
            // (slot p = number of hitting rays with a smaller index); each lane then counts the slots with a smaller key,
            // four slots per trip of a wave-uniform loop bounded by ceil(#hits / 4) <= 8.  Unused slots hold 0xFFFFFFFF.
            const int p = __popc(myL & below);
            L.tk[lane] = 0xFFFFFFFFu;
            WSYNC();
            if (isL) L.tk[hi_half * 32 + p] = kr;
            WSYNC();
            const int c_lo = z_lo ? 0 : __popc(lo), c_hi = z_hi ? 0 : __popc(hi);
            const int cmax = c_lo > c_hi ? c_lo : c_hi;
            int rlt = 0;
    #pragma unroll 1
            for (int cb = 0; cb < cmax; cb += 4) {
              const uint4 ks = *reinterpret_cast<const uint4*>(&L.tk[hi_half * 32 + cb]);
              rlt += (ks.x < kr ? 1 : 0) + (ks.y < kr ? 1 : 0) + (ks.z < kr ? 1 : 0) + (ks.w < kr ? 1 : 0);
            }
            // equal keys among the hits (two rays with bit-identical alpha) would need the index tie-break: they show up as
            // two lanes claiming the same strict rank.  Detect through a second pass over the slot list (claim slot rlt with
            // p, read it back) and only then add the tie-break term; practically never taken.
            WSYNC();
            if (isL) L.tk[hi_half * 32 + rlt] = (uint32_t)p;
            WSYNC();
            const bool lost = isL && (L.tk[hi_half * 32 + rlt] != (uint32_t)p);
            if (__builtin_amdgcn_ballot_w64(lost) != 0ull) {
              WSYNC();
              L.tk[lane] = 0xFFFFFFFFu;
              WSYNC();
              if (isL) L.tk[hi_half * 32 + p] = kr;
              WSYNC();
              rlt = 0;
    #pragma unroll 1
              for (int sl = 0; sl < cmax; ++sl) {
                const uint32_t kj = L.tk[hi_half * 32 + sl];
                rlt += (kj < kr || (kj == kr && sl < p)) ? 1 : 0;
              }
              WSYNC();
            }
            rank = allz ? r : (isL ? rlt : rank_other);
          }
          if (rank < K) {
            const float hx = x1 + (x2 - x1) * ar, hy = y1 + (y2 - y1) * ar;
            const int hq = i * K + rank, node = 2 * NA + hq;
            L.hits[hq * 2] = hx; L.hits[hq * 2 + 1] = hy;
            if (has_graph) { L.nodes[node * ND] = hx; L.nodes[node * ND + 1] = hy; }
          }
          PHASE("P3_end", 12 + it * 3 + 2);
        }
      }
    } else {
      // materialise-only: the hit points are given
      for (int q = lane; q < NA * K; q += 64) {
        const float hx = L.hits[q * 2], hy = L.hits[q * 2 + 1];
        const int node = 2 * NA + q;
        if (has_graph) { L.nodes[node * ND] = hx; L.nodes[node * ND + 1] = hy; }
      }
    }
    WSYNC();
    PHASE("P4_compact", 7);
    // ---- P4: late compact outputs ----
    if (a.next_hits != nullptr) {
      float4* o4 = reinterpret_cast<float4*>(a.next_hits + (size_t)b * NA * K * 2);
      for (int i = lane; i < NA * K * 2 / 4; i += 64) o4[i] = reinterpret_cast<const float4*>(L.hits)[i];
    }
    if (!has_graph) continue;
    PHASE("P5_graph", 8);
    // ---- P5: padded GraphsTuple (lidar_env/base.py:227-271, lidar_spread.py:57-96, lidar_target.py:57-96, graph.py:35-44,212-247)
    {
      float4* edges = reinterpret_cast<float4*>(a.g.edges) + (size_t)b * E;
      int32_t* recv = a.g.receivers + (size_t)b * E;
      int32_t* send = a.g.senders + (size_t)b * E;
#pragma unroll
      for (int q0 = 0; q0 < NA * K; q0 += 64) {      // agent-hit block
        const int q = q0 + lane;
        if (q < NA * K) {
          const int i = q / K;
          const float lx = L.next[i * SD] - L.hits[q * 2], ly = L.next[i * SD + 1] - L.hits[q * 2 + 1];
          const bool mask = lx * lx + ly * ly < a.thr2_lidar;      // sqrt(s) < comm_radius - 0.1, as above
          edges[NA * NA + NA * GS + q] = make_float4(lx, ly, 0.0f, 0.0f);
          recv[NA * NA + NA * GS + q] = mask ? i : PAD;
          send[NA * NA + NA * GS + q] = mask ? 2 * NA + q : PAD;
        }
      }
      float* nodes = a.g.nodes + (size_t)b * N * ND;
#pragma unroll
      for (int q0 = 0; q0 < N * ND; q0 += 64) {
        const int q = q0 + lane;
        if (q < N * ND) nodes[q] = L.nodes[q];
      }
      // states [N, SD] = the leading SD columns of the node rows, pad row -1 (lidar_env/base.py:260-264, graph.py:217-218)
      if constexpr (SD == 4) {
        float4* st4 = reinterpret_cast<float4*>(a.g.states + (size_t)b * N * SD);
#pragma unroll
        for (int q0 = 0; q0 < N; q0 += 64) {
          const int node = q0 + lane;
          if (node < N) {
            const float* row = L.nodes + node * ND;
            st4[node] = (node == PAD) ? make_float4(-1.0f, -1.0f, -1.0f, -1.0f) : make_float4(row[0], row[1], row[2], row[3]);
          }
        }
      } else {
        float* st = a.g.states + (size_t)b * N * SD;
#pragma unroll
        for (int q0 = 0; q0 < N * SD; q0 += 64) {
          const int q = q0 + lane;
          if (q < N * SD) {
            const int node = q / SD, d = q - node * SD;
            st[q] = (node == PAD) ? -1.0f : L.nodes[node * ND + d];
          }
        }
      }
    }
    WSYNC();   // the next env patches the images only after this env's reads of them were issued
    PHASE("env_end", 9);
#ifdef DGPPO_STAMPS
    if (blockIdx.x == 0 && threadIdx.x == 0 && stamp_on) g_wstamps[41] = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0 && gw < 8192) g_wspan[3 * gw + 2] = __builtin_amdgcn_s_memtime();
#endif
  }
}

template <int SD, bool SPREAD, int NA, int NO>
bool launch_inst(const StepArgs& a, hipStream_t s) {
  constexpr size_t per_wave = sizeof(WaveLds<SD, SPREAD, NA, NO, true>);        // the larger of the two layouts
  // waves per workgroup: as many independent waves as keep several workgroups resident in the 160 KiB of a CU
  constexpr int WPB = (per_wave * 4 <= 40 * 1024) ? 4 : ((per_wave * 2 <= 52 * 1024) ? 2 : 1);
  static_assert(per_wave * WPB <= 64 * 1024, "LDS slab too large");
  constexpr size_t smem_g = per_wave * WPB, smem_c = sizeof(WaveLds<SD, SPREAD, NA, NO, false>) * WPB;
  // persistent grid = what is actually resident (registers, LDS and wave slots together): ask the runtime once per
  // instantiation; a larger grid would leave a second, partial round of workgroups behind the first
  static int per_cu_cached[3][2] = {{0, 0}, {0, 0}, {0, 0}};
  int& per_cu = per_cu_cached[a.mode][a.has_graph ? 1 : 0];
  if (per_cu == 0) {
    int n = 0;
    hipError_t e = hipErrorUnknown;
#define OCC(M_, G_) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, lidar_wave_kernel<SD, SPREAD, NA, NO, WPB, M_, G_>, WPB * 64, (G_) ? smem_g : smem_c)
    if (a.mode == MODE_STEP) { if (a.has_graph) OCC(MODE_STEP, true); else OCC(MODE_STEP, false); }
    else if (a.mode == MODE_SENSE) { if (a.has_graph) OCC(MODE_SENSE, true); else OCC(MODE_SENSE, false); }
    else OCC(MODE_GRAPH, true);
#undef OCC
    per_cu = (e == hipSuccess && n > 0) ? n : 1;
  }
  int n_cu = 256;
  {
    static int cached_cu = 0;
    if (cached_cu == 0) {
      int dev = 0, v = 0;
      if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
        cached_cu = v;
      else
        cached_cu = 256;
    }
    n_cu = cached_cu;
  }
  const int max_blocks = n_cu * per_cu;
  int blocks = (a.B + WPB - 1) / WPB;
  if (blocks > max_blocks) blocks = max_blocks;
  const char* wpe = getenv("DGPPO_WAVE_ENVS");    // tuning knob: minimum envs per wave (fewer, longer-lived waves)
  if (wpe && atoi(wpe) > 1) {
    const int want = (a.B + WPB * atoi(wpe) - 1) / (WPB * atoi(wpe));
    if (want >= 1 && want < blocks) blocks = want;
  }
  const char* gpe = getenv("DGPPO_WAVE_GRID_ENVS");   // tuning knob: exactly this many envs per wave, grid NOT capped at the resident set
  if (gpe && atoi(gpe) >= 1) blocks = (a.B + WPB * atoi(gpe) - 1) / (WPB * atoi(gpe));
  const dim3 g(blocks), t(WPB * 64);
#define LAUNCH(M_, G_) hipLaunchKernelGGL((lidar_wave_kernel<SD, SPREAD, NA, NO, WPB, M_, G_>), g, t, (G_) ? smem_g : smem_c, s, a)
  if (a.mode == MODE_STEP) { if (a.has_graph) LAUNCH(MODE_STEP, true); else LAUNCH(MODE_STEP, false); }
  else if (a.mode == MODE_SENSE) { if (a.has_graph) LAUNCH(MODE_SENSE, true); else LAUNCH(MODE_SENSE, false); }
  else LAUNCH(MODE_GRAPH, true);
#undef LAUNCH
  return true;
}

}  // namespace

#ifdef DGPPO_STAMPS
extern "C" int32_t dgppo_debug_wave_stamps(unsigned long long* out) {
  return (int32_t)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wstamps), sizeof(unsigned long long) * 64);
}
extern "C" int32_t dgppo_debug_wave_spans(unsigned long long* out) {
  return (int32_t)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wspan), sizeof(unsigned long long) * 3 * 8192);
}
#endif

bool launch_lidar_wave(const StepArgs& a, hipStream_t s) {
  const dgppo_env_cfg& c = a.cfg;
  if (!cfg_is_lidar(c) || !cfg_is_base_kind(c) || c.n_rays != 32 || c.top_k != 8 || c.n_obs < 1) return false;
  if (getenv("DGPPO_NO_WAVE_ENV_KERNEL")) return false;
  if (!(c.eye_offset >= c.comm_radius)) return false;          // the diagonal of the agent-agent block must be masked
  // 16-byte staging loads need 16-byte aligned bases (torch allocations are; sliced views may not be)
  auto al16 = [](const void* p) { return p == nullptr || (((uintptr_t)p) & 15) == 0; };
  if (!(al16(a.agent) && al16(a.goal) && al16(a.obst) && al16(a.hits) && al16(a.next_agent) && al16(a.next_hits) &&
        al16(a.action)))
    return false;
  if (a.has_graph && !(al16(a.g.edges) && al16(a.g.states))) return false;
  const bool spread = cfg_is_spread(c);
  const int n = c.n_agents, no = c.n_obs, sd = c.state_dim;
#define TRY(SD_, SP_, NA_, NO_) if (sd == SD_ && spread == SP_ && n == NA_ && no == NO_) return launch_inst<SD_, SP_, NA_, NO_>(a, s)
  TRY(4, true, 8, 3);      // BASELINE configs 3 / 4: LidarSpread n = 8, obs = 3
  TRY(4, false, 8, 3);     // LidarTarget n = 8, obs = 3
  TRY(5, false, 16, 8);    // BASELINE config 5: LidarBicycleTarget n = 16, obs = 8
  TRY(4, true, 16, 8);
  TRY(4, true, 4, 2);
  TRY(4, false, 4, 2);
  TRY(5, false, 4, 3);
#undef TRY
  return false;
}
