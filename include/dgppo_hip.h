/*
 * dgppo_hip.h — C ABI of libdgppo_hip.so, the MI355X (gfx950) implementation of
 * DGPPO's data-parallel hot path.
 *
 * The reference (syzhang092218-source/dgppo) is pure Python/JAX and has NO FFI
 * of its own; the boundary it exposes is the Python API (dgppo.env / dgppo.algo /
 * dgppo.trainer).  This header is therefore the C ABI *underneath* that Python
 * API: every entry point names the reference function(s) (file:line relative to
 * /root/reference) whose arithmetic it replaces.  INTEGRATION.md shows the
 * ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (a torch.Tensor kept
 *     alive by Python); the library never allocates persistent memory, keeps no
 *     global mutable state apart from the thread-local error string, and never
 *     synchronises: all work is enqueued on the caller-supplied hipStream_t
 *     (passed as void*; NULL = the null stream);
 *   - all floating point is IEEE fp32, all indices int32, row-major, contiguous;
 *   - return value: 0 ok, <0 bad argument (see dgppo_last_error()), >0 hipError_t;
 *   - shapes are passed explicitly and validated on the host before any launch.
 */
#ifndef DGPPO_HIP_H
#define DGPPO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DGPPO_ABI_VERSION 3

/* environment kinds — dgppo/env/__init__.py:9-23 (registered ids on the hot path) */
enum {
  DGPPO_ENV_LIDAR_SPREAD = 0,         /* dgppo/env/lidar_env/lidar_spread.py */
  DGPPO_ENV_LIDAR_TARGET = 1,         /* dgppo/env/lidar_env/lidar_target.py */
  DGPPO_ENV_LIDAR_BICYCLE_TARGET = 2, /* dgppo/env/lidar_env/lidar_bicycle_target.py */
  DGPPO_ENV_MPE_SPREAD = 3,           /* dgppo/env/mpe/mpe_spread.py */
  DGPPO_ENV_MPE_TARGET = 4,           /* dgppo/env/mpe/mpe_target.py */
  /* task variants: same step / graph kernels, different reset, reward goals, goal-node count */
  DGPPO_ENV_LIDAR_LINE = 5,           /* dgppo/env/lidar_env/lidar_line.py : 2 landmark nodes, goals on the segment */
  DGPPO_ENV_MPE_LINE = 6,             /* dgppo/env/mpe/mpe_line.py */
  DGPPO_ENV_MPE_FORMATION = 7,        /* dgppo/env/mpe/mpe_formation.py : 1 landmark node, goals on a circle */
  DGPPO_ENV_MPE_CORRIDOR = 8,         /* dgppo/env/mpe/mpe_corridor.py : two fixed discs, y limit 2 * area */
  DGPPO_ENV_MPE_CONNECT_SPREAD = 9    /* dgppo/env/mpe/mpe_connect_spread.py : third (connectivity) cost */
};
/* dgppo_env_cfg.reward_goals: how the n positions the reward measures against follow from the goal nodes */
enum {
  DGPPO_GOALS_NODES = 0,         /* the goal nodes themselves                                   lidar_spread.py:35-52   */
  DGPPO_GOALS_LINE = 1,          /* l0 + i (l1 - l0) / (n - 1), i = 0..n-1                      lidar_line.py:131-136   */
  DGPPO_GOALS_LINE_INTERIOR = 2, /* l0 + (i + 1) (l1 - l0) / (n + 1)  (MPELine, n <= 3)         mpe_line.py:124-133     */
  DGPPO_GOALS_CIRCLE = 3         /* landmark + comm_radius [cos, sin](2 pi i / n)               mpe_formation.py:94-98  */
};

/* Obstacle record of the LiDAR envs: 16 floats per rectangle
 * (dgppo/env/obstacle.py:30-56  Rectangle(type, center, width, height, theta, points)):
 *   [0..1] center  [2] width  [3] height  [4] theta  [5] cos(theta)  [6] sin(theta)  [7] 0
 *   [8..15] points[4][2]
 * MPE obstacles are plain state rows [ox, oy, 0, 0] (dgppo/env/mpe/base.py:121).        */
#define DGPPO_RECT_STRIDE 16

/* Static description of one environment family + PARAMS
 * (dgppo/env/lidar_env/lidar_spread.py:13-22, dgppo/env/mpe/mpe_spread.py:12-19,
 *  dgppo/env/__init__.py:29-53).  Thresholds that the reference forms from Python
 * doubles and then rounds to fp32 are passed pre-rounded so host and device agree. */
typedef struct dgppo_env_cfg {
  int32_t kind;        /* DGPPO_ENV_* */
  int32_t n_agents;    /* n */
  int32_t n_goals;     /* goal NODES of the graph: n, 2 (Line: landmarks) or 1 (Formation) */
  int32_t n_obs;       /* rectangles (LiDAR) or discs (MPE); may be 0 */
  int32_t n_rays;      /* R (LiDAR only) */
  int32_t top_k;       /* k = top_k_rays (LiDAR only) */
  int32_t state_dim;   /* 4, or 5 for the bicycle */
  int32_t node_dim;    /* state_dim + 3 */
  float area_size;
  float dt;
  float car_radius;
  float comm_radius;
  float obs_radius;        /* MPE only */
  float dist2goal;
  float two_car_radius;    /* fp32(2 * car_radius)                       lidar_env/base.py:188 */
  float lidar_mask_radius; /* fp32(comm_radius - 1e-1)                   lidar_spread.py:88   */
  float eye_offset;        /* fp32(comm_radius + 1)                      lidar_spread.py:64   */
  float car_plus_obs;      /* fp32(car_radius + obs_radius)              mpe/base.py:181      */
  float vel_limit;         /* 0.5 (LiDAR) or 1.0 (MPE)                   state_lim()          */
  float reset_min_dist;    /* fp32(2.2*car_radius) LiDAR, fp32(2*car_radius) MPE, fp32(2.3*car_radius) ConnectSpread */
  /* ---- task variants (ABI 3); defaults reproduce the five base kinds ---- */
  int32_t reward_goals;    /* DGPPO_GOALS_*                                                                          */
  int32_t n_cost;          /* 2, or 3 with the connectivity cost                   mpe_connect_spread.py:46-52,115-117 */
  float obs_mask_radius;   /* MPE agent-obstacle edge mask: comm_radius, or fp32(100*comm_radius)  mpe_corridor.py:93   */
  float y_limit;           /* upper state limit in y: area_size or fp32(2*area_size)               mpe_corridor.py:62-65 */
  float connect_radius;    /* mpe_connect_spread.py:24                                                                */
  float reset_side_y;      /* height of the box agents / goals are sampled in at reset             mpe_corridor.py:50   */
  float goal_shift_y;      /* added to the sampled goals' y                                        mpe_corridor.py:52   */
  float line_min_dist;     /* minimum landmark separation                                          mpe_line.py:49-52    */
} dgppo_env_cfg;

/* Optional materialised GraphsTuple (dgppo/utils/graph.py:47-86, GetGraph.to_padded
 * :212-247).  Any pointer group may be NULL as a whole (all-or-nothing).            */
typedef struct dgppo_graph_out {
  float*   nodes;      /* [B, N, node_dim]   */
  float*   edges;      /* [B, E, 4]          */
  float*   states;     /* [B, N, state_dim]  */
  int32_t* receivers;  /* [B, E]             */
  int32_t* senders;    /* [B, E]             */
  int32_t* node_type;  /* [B, N]             */
  int32_t* n_node;     /* [B]                */
  int32_t* n_edge;     /* [B]                */
} dgppo_graph_out;

int32_t     dgppo_abi_version(void);
const char* dgppo_last_error(void);

/* number of nodes N (incl. pad) / edges E of the padded graph — graph.py:212-247 */
int32_t dgppo_env_num_nodes(const dgppo_env_cfg* cfg);
int32_t dgppo_env_num_edges(const dgppo_env_cfg* cfg);

/* ---- environment ---------------------------------------------------------------- */

/* One batched env.step: replaces LidarEnv.step (dgppo/env/lidar_env/base.py:151-174),
 * MPE.step (dgppo/env/mpe/base.py:137-162) and everything they call:
 *   clip_action/clip_state (env/base.py:80-86), agent_step_euler (lidar_env/base.py:142-149,
 *   lidar_bicycle_target.py:92-111, mpe/base.py:129-135), get_lidar/raytracing/top-k
 *   (env/utils.py:49-55,115-136; obstacle.py:62-105), get_reward (lidar_spread.py:35-52,
 *   lidar_target.py:35-52, mpe_spread.py:32-49, mpe_target.py:32-49), get_cost
 *   (lidar_env/base.py:180-207, mpe/base.py:164-191), edge_blocks + get_graph + to_padded.
 *
 *   agent      [B, n, sd]   state at t
 *   action     [B, n, 2]    raw action (clipped inside).  NULL => "sense only": no dynamics,
 *                           no reward/cost; next_agent := agent (used by reset to build graph_0)
 *   goal       [B, ng, sd]
 *   obst       LiDAR: [B, n_obs, 16] rectangle records; MPE: [B, n_obs, sd]; may be NULL iff n_obs==0
 *   hits       LiDAR: [B, n, k, 2] hit points of the graph at t (obstacle cost reads them,
 *                           lidar_env/base.py:194-197); NULL for MPE / n_obs==0 / sense-only
 *   ray_cos/ray_sin [R]     cos/sin(linspace(-pi, pi-2pi/R, R)) (env/utils.py:51), fp32
 *   next_agent [B, n, sd]   out
 *   next_hits  [B, n, k, 2] out (LiDAR)
 *   reward     [B]          out (reward of step t, on the pre-step graph)
 *   cost       [B, n, 2]    out
 *   gout       optional materialised graph at t+1                                        */
int32_t dgppo_env_step(const dgppo_env_cfg* cfg,
                       const float* agent, const float* action, const float* goal,
                       const float* obst, const float* hits,
                       const float* ray_cos, const float* ray_sin,
                       float* next_agent, float* next_hits, float* reward, float* cost,
                       const dgppo_graph_out* gout, int32_t B, void* stream);

/* Materialise the GraphsTuple of a stored compact record (lazy rollout.graph view):
 * same arithmetic as get_graph (lidar_env/base.py:227-271, mpe/base.py:211-241).     */
int32_t dgppo_graph_materialize(const dgppo_env_cfg* cfg,
                                const float* agent, const float* goal,
                                const float* obst, const float* hits,
                                const dgppo_graph_out* gout, int32_t B, void* stream);

/* Batched reset: replaces LidarEnv.reset (lidar_env/base.py:89-124), LidarBicycleTarget.reset
 * (lidar_bicycle_target.py:60-90), MPE.reset (mpe/base.py:81-127), get_node_goal_rng
 * (env/utils.py:139-244), Rectangle.create (obstacle.py:39-56).  Philox-4x32-10 counter RNG
 * keyed by seeds[b] (the JAX threefry stream cannot be reproduced).
 *   seeds [B] uint64;  out: agent [B,n,sd], goal [B,ng,sd], obst (layout as above).      */
int32_t dgppo_env_reset(const dgppo_env_cfg* cfg, const uint64_t* seeds,
                        float* agent, float* goal, float* obst, int32_t B, void* stream);
/* The same with a failure counter.  The reference's rejection loops (env/utils.py:139-244 `while_loop`s, mpe_connect_spread.py
 * :50-107) are unbounded; the kernels bound every loop so that all threads finish, and an env whose bound ran out holds an
 * INVALID scene.  *n_failed (DEVICE int32, caller-zeroed, may be NULL) is incremented once per such env; the caller reads it
 * at its next host sync and must not train on the batch when it is non-zero.  Both entry points also reject, on the host
 * and before launching, densities at which the placement cannot succeed (negative return).                              */
int32_t dgppo_env_reset_checked(const dgppo_env_cfg* cfg, const uint64_t* seeds, float* agent, float* goal, float* obst,
                                int32_t* n_failed, int32_t B, void* stream);

/* Standard-normal noise: Philox-4x32-10 + Box-Muller, out[i] for i<n_elem; replaces the
 * jax.random draw inside dist.sample(seed=key) (algo/module/policy.py:196-203).         */
int32_t dgppo_randn(uint64_t seed, uint64_t offset, float* out, int64_t n_elem, void* stream);
/* A column window of the same stream laid out as rows: out[r, c] (dense [rows, row_len]) = element r * global_row_len +
 * col_offset + c of dgppo_randn(seed, 0, ...).  Data-parallel rollouts: the sampling noise of step t is a row of
 * global_row_len = world * B_local * n * 2 normals and a rank fills the window of its envs, so that the union of the ranks'
 * rollouts is the single-device rollout of the global batch (jax.random.split(key, B) of dgppo/algo/informarl.py:254-256 is
 * likewise a function of the global env index).                                                                          */
int32_t dgppo_randn_rows(uint64_t seed, float* out, int64_t rows, int64_t row_len, int64_t global_row_len,
                         int64_t col_offset, void* stream);

/* ---- networks: building blocks ------------------------------------------------------------ */
/* All matrices row-major fp32; `ld*` = leading dimension in floats.                              */

/* Y[M,N] = act(X[M,K] W[K,N] + bias) on the fp32 matrix cores; act 0 none / 1 relu; accumulate: Y += ...;
 * trans_w: use W^T (W stored [N,K]) — the input-gradient of a Dense.  relu_mask (optional, [M,N], leading dimension ldm):
 * Y = (relu_mask > 0) ? result : 0 applied last — the backward of the ReLU whose OUTPUT relu_mask is, fused into the
 * kernel that finishes that gradient (jax.grad of nn.relu in dgppo/nn/gnn.py:39, mlp.py:29).  Replaces flax nn.Dense as
 * used by dgppo/nn/mlp.py:19-22, dgppo/nn/gnn.py:86-110, dgppo/nn/rnn.py:19-21, algo/module/policy.py:67-70, value.py:41,76. */
int32_t dgppo_dense_fwd(const float* X, int32_t ldx, const float* W, int32_t ldw, const float* bias, float* Y,
                        int32_t ldy, int32_t M, int32_t K, int32_t N, int32_t act, int32_t accumulate,
                        int32_t trans_w, const float* relu_mask, int32_t ldm, void* stream);
/* dW[K,N] += X^T dY ; db[N] += colsum(dY) (db may be NULL): the weight-gradient of a Dense
 * (jax.grad at dgppo/algo/informarl.py:377,440 ; dgppo/algo/dgppo.py:316).                          */
int32_t dgppo_dense_bwd_w(const float* X, int32_t ldx, const float* dY, int32_t ldy, float* dW, int32_t ldw,
                          float* db, int32_t M, int32_t K, int32_t N, float* workspace, int64_t workspace_bytes,
                          void* stream);
/* The MLP trunk and the GRU input projection as PPOPolicy / ValueNet compose them (dgppo/algo/module/policy.py:191-212,
 * value.py:58-80) in one launch: y1 = relu(LN(X W1 + b1)), y2 = relu(LN(y1 W2 + b2)) (MLP of dgppo/nn/mlp.py:17-29, hidden
 * 64, LayerNorm scale g / bias be, eps 1e-6), gi = y2 Wi + bi [M,192] (the input half of flax GRUCell, dgppo/nn/rnn.py:14-30,
 * gates r|z|n).  X [M, >= 64] with leading dimension ldx.  p1/y1/st1/p2/y2/st2 (pre-LN [M,64], post-ReLU [M,64], LN
 * statistics [M,2] per layer) are what the backward needs: all six or none (inference).                          */
int32_t dgppo_mlp_gi_fwd(const float* X, int32_t ldx, const float* W1, const float* b1, const float* g1, const float* be1,
                         const float* W2, const float* b2, const float* g2, const float* be2, const float* Wi,
                         const float* bi, float* p1, float* y1, float* st1, float* p2, float* y2, float* st2, float* gi,
                         int32_t M, void* stream);
/* Backward of that chain with respect to its activations, in one launch (jax.grad through dgppo/nn/mlp.py:17-29 and the
 * GRUCell input Dense of dgppo/nn/rnn.py:14-30, as taken by dgppo/algo/informarl.py:377,440 / dgppo/algo/dgppo.py:316):
 * dpre2 = LNReLU'(p2, y2, st2, g2; dgi Wi^T), dpre1 = LNReLU'(p1, y1, st1, g1; dpre2 W2^T), dx = dpre1 W1^T, where
 * LNReLU'(p, y, st, g; dy) = rstd (dxh - mean(dxh) - xhat mean(dxh xhat)), dxh = dy (y > 0) g.  relu_mask [M, ldm] (may be
 * NULL): dx = (mask > 0) ? dx : 0.  dpre2 / dpre1 [M,64] are outputs (the weight gradients dW2 = y1^T dpre2, dW1 = X^T dpre1
 * read them); dg2/db2/dg1/db1 [64] += the LayerNorm scale / bias gradients.  dgi [M,192] 16-byte aligned.          */
int32_t dgppo_mlp_gi_bwd(const float* dgi, const float* Wi, const float* W2, const float* W1, const float* g2,
                         const float* g1, const float* p2, const float* y2, const float* st2, const float* p1,
                         const float* y1, const float* st1, const float* relu_mask, int32_t ldm, float* dpre2, float* dpre1,
                         float* dx, int32_t lddx, float* dg2, float* db2, float* dg1, float* db1, int32_t M, void* stream);
/* The same weight gradient with its second stage DEFERRED: only the partial-slab kernel is launched; *pending describes the
 * reduction that is still owed (pending->pending = 0 when the call needed none).  The caller owns the descriptors and the
 * workspace REGION of every deferred call until it has flushed them with dgppo_dense_bwd_w_reduce_batch (one launch per
 * DGPPO_REDUCE_BATCH reductions, on the same stream).  A backward pass has ~12 weight gradients.                    */
#define DGPPO_REDUCE_BATCH 16
typedef struct dgppo_reduce_desc {
  const float* part;     /* partial slabs [slabs][part_stride] inside the caller's workspace region */
  float* dW;             /* [K, N] with leading dimension ldw: dW += sum of the slabs               */
  float* db;             /* [N] or NULL                                                              */
  int32_t part_stride, slabs, ldw, K, N;
  int32_t pending;       /* 1: this reduction is owed                                                */
} dgppo_reduce_desc;
int32_t dgppo_dense_bwd_w_deferred(const float* X, int32_t ldx, const float* dY, int32_t ldy, float* dW, int32_t ldw,
                                   float* db, int32_t M, int32_t K, int32_t N, float* workspace, int64_t workspace_bytes,
                                   dgppo_reduce_desc* pending, void* stream);
int32_t dgppo_dense_bwd_w_reduce_batch(const dgppo_reduce_desc* descs, int32_t n, void* stream);
/* One GRU step (T = 1) fused with the output Dense layer(s) on the same rows — the tail of a rollout step:
 * h' = GRUCell(gi, h0) (dgppo/nn/rnn.py:14-30, gi = x W_i + b_i from dgppo_mlp_gi_fwd), then either the policy head
 * u = h' W1 + b1 [64], out = u W2 + b2 (PolicyNet.head Dense -> TanhNormal Dense, dgppo/algo/module/policy.py:62-74; pass
 * W2/b2) or a value head out = h' W1 + b1 (dgppo/algo/module/value.py:41,76; W2 = b2 = NULL).  n_out <= 16.  hprev [M,64],
 * gates [M,256] (r|z|n|hn_lin) and u [M,64] are the activations the backward needs (each may be NULL).       */
int32_t dgppo_gru1_head_fwd(const float* gi, const float* Wh, const float* bhn, const float* h0, const float* W1,
                            const float* b1, const float* W2, const float* b2, float* hs, float* hprev, float* gates,
                            float* u, float* out, int32_t M, int32_t n_out, void* stream);
/* Scratch for dgppo_dense_bwd_w's two-stage reduction (per-workgroup partial sums, then one reduce kernel): the caller
 * owns it, like every other buffer (16-byte aligned device memory, reusable by consecutive calls on one stream).  The
 * returned size lets every resident workgroup keep its own slab; a smaller buffer shrinks the grid, NULL / 0 falls back
 * to atomicAdd into dW (correct, slower).  No reference counterpart (XLA owns its scratch allocations).      */
int64_t dgppo_dense_bwd_w_workspace_bytes(int32_t K, int32_t N);

/* Compact record -> per-graph dense features for the GNN: agent node rows Xa [G*n,Fp], other node rows
 * Xo [G*(Ns-n),Fp], per-(agent,slot) edge features [G*n,S,4] and masks [G*n,S] (1/0).  Graph g = e*n_time + t reads
 * agent + env*agent_se + t*agent_st (env = env_ids ? env_ids[e] : e), likewise hits.  Same arithmetic as
 * get_graph/edge_blocks (lidar_env/base.py:227-271, lidar_spread.py:57-96, lidar_target.py:57-96, mpe twins).         */
int32_t dgppo_graph_feats(const dgppo_env_cfg* cfg, const float* agent, int64_t agent_se, int64_t agent_st,
                          const float* goal, const float* obst, const float* hits, int64_t hits_se, int64_t hits_st,
                          const int32_t* env_ids, int32_t n_env, int32_t n_time, float* Xa, float* Xo, float* efeat,
                          float* emask, int32_t Fp, void* stream);

/* GraphTransformer attention in fixed-fan-in form (dgppo/nn/gnn.py:85-117; jraph.segment_softmax/segment_sum):
 * qt [G*n,H*F] = x_i Mcat + cvec (a Dense), logits = qt . x_sender, masked softmax over the S slots of each agent,
 * zcat [G*n,Kp] = [x_i | per head: sum a x_s (F), sum a e (4) | 1 | 0...], attn [G*n,S,H] saved for backward (NULL in
 * inference: the weights are not written).                                                                          */
int32_t dgppo_attn_fwd(const dgppo_env_cfg* cfg, int32_t F, int32_t H, int32_t Kp, const float* qt, const float* Xa,
                       const float* Xo, const float* efeat, const float* emask, float* zcat, float* attn, int32_t G,
                       void* stream);
/* backward of the above: dqt [G*n,H*F]; dXa [G*n,F] / dXo [G*(Ns-n),F] written (not accumulated) when non-NULL.
 * relu_xo != 0: dXo *= (Xo > 0) — Xo is then the ReLU output of the previous layer's node update, and its gradient
 * needs no separate pass (dXa still receives the dqt Mcat^T term from dgppo_dense_fwd, which applies the mask).        */
int32_t dgppo_attn_bwd(const dgppo_env_cfg* cfg, int32_t F, int32_t H, int32_t Kp, const float* dzcat,
                       const float* attn, const float* qt, const float* Xa, const float* Xo, const float* efeat,
                       float* dqt, float* dXa, float* dXo, int32_t relu_xo, int32_t G, void* stream);
/* The same layer (F = 32) with the sender rows of the nodes WITHOUT incoming edges (goals, LiDAR hits, obstacles) recomputed
 * inside the kernel instead of read: Xo = relu(Xo_raw Wo + bo), Xo_raw [G*(Ns-n),8] their padded raw features, Wo [8,ldwo] the
 * first 8 rows of the previous layer's update weight (Wout of dgppo_gnn_prep), bo [32] its bias — what the reference's
 * GraphTransformer layer produces for a node with aggr = 0 (dgppo/nn/gnn.py:109-111) before the next layer reads it as a
 * sender (gnn.py:85-117).  Saves the 9 KB per graph of materialised rows in the forward and again in the backward.
 * dgppo_attn_xo_supported: 1 if the topology has such a kernel (8 LiDAR hits per agent or no private nodes, <= 32 shared
 * nodes, H <= 4), else 0 -> materialise Xo with dgppo_dense_fwd and call dgppo_attn_fwd / dgppo_attn_bwd.
 * dXo of the backward is the gradient w.r.t. the RECOMPUTED rows (relu_xo != 0: already multiplied by relu'), i.e. exactly
 * what dgppo_dense_bwd_w needs for dWo / dbo.                                                                            */
int32_t dgppo_attn_xo_supported(const dgppo_env_cfg* cfg, int32_t F, int32_t H, int32_t Kp);
int32_t dgppo_attn_fwd_xo(const dgppo_env_cfg* cfg, int32_t F, int32_t H, int32_t Kp, const float* qt, const float* Xa,
                          const float* Xo_raw, const float* Wo, int32_t ldwo, const float* bo, const float* efeat,
                          const float* emask, float* zcat, float* attn, int32_t G, void* stream);
int32_t dgppo_attn_bwd_xo(const dgppo_env_cfg* cfg, int32_t F, int32_t H, int32_t Kp, const float* dzcat,
                          const float* attn, const float* qt, const float* Xa, const float* Xo_raw, const float* Wo,
                          int32_t ldwo, const float* bo, const float* efeat, float* dqt, float* dXa, float* dXo,
                          int32_t relu_xo, int32_t G, void* stream);
/* dgppo_attn_bwd_xo that CONSUMES the gradient of the recomputed rows instead of writing it (9 KB per graph that
 * dgppo_dense_bwd_w would read again): dWo [8,lddwo] += Xo_raw^T dpre, dbo [32] += colsum(dpre), dpre = relu'(Xo) * dXo — the
 * weight gradient jax.grad gives the previous layer's update Dense for the nodes without incoming edges (gnn.py:109-111).
 * workspace: dgppo_attn_xo_workspace_bytes(G) bytes, caller-owned, 16-byte aligned (per-graph partial sums, reduced by a
 * second launch on the same stream).                                                                                      */
int64_t dgppo_attn_xo_workspace_bytes(int32_t G);
int32_t dgppo_attn_bwd_xo_dw(const dgppo_env_cfg* cfg, int32_t F, int32_t H, int32_t Kp, const float* dzcat,
                             const float* attn, const float* qt, const float* Xa, const float* Xo_raw, const float* Wo,
                             int32_t ldwo, const float* bo, const float* efeat, float* dqt, float* dXa, float* dWo,
                             int32_t lddwo, float* dbo, float* workspace, int64_t workspace_bytes, int32_t G, void* stream);

/* GraphTransformer parameters (flax Dense_0..4 = q,k,v,e,u; gnn.py:86-110) -> Mcat [Fp,H*Fp], cvec [H*Fp],
 * Wout [Kp,D] used by dgppo_attn_* and the surrounding Denses; and the adjoint map (accumulates into d*).           */
int32_t dgppo_gnn_prep(const float* Wq, const float* bq, const float* Wk, const float* Wv, const float* bv,
                       const float* We, const float* Wu, float* Mcat, float* cvec, float* Wout, int32_t F, int32_t Fp,
                       int32_t D, int32_t H, int32_t Kp, void* stream);
int32_t dgppo_gnn_unprep(const float* dMcat, const float* dcvec, const float* dWout, const float* Wq, const float* bq,
                         const float* Wk, float* dWq, float* dbq, float* dWk, float* dWv, float* dbv, float* dWe,
                         float* dWu, int32_t F, int32_t Fp, int32_t D, int32_t H, int32_t Kp, void* stream);

/* y = relu(LayerNorm_64(x)) (dgppo/nn/mlp.py:27-29; flax LayerNorm eps 1e-6); stats [M,2] = mean, rstd.          */
int32_t dgppo_ln_relu_fwd(const float* x, const float* gamma, const float* beta, float* y, float* stats, int32_t M,
                          void* stream);
int32_t dgppo_ln_relu_bwd(const float* x, const float* y, const float* stats, const float* gamma, const float* dy,
                          float* dx, float* dgamma, float* dbeta, int32_t M, void* stream);

/* GRU scan (dgppo/nn/rnn.py:14-30, flax GRUCell, 64 features).  gi [rows,192] = x Wi + bi precomputed; Wh [64,192]
 * (r|z|n); sequence s at step tau lives in row ((s/n_inner)*T + tau)*n_inner + s%n_inner.  h0 [n_seq,64] or NULL.
 * hs [rows,64]; hprev [rows,64] and gates [rows,256] are saved for the backward when non-NULL.                      */
int32_t dgppo_gru_fwd(const float* gi, const float* Wh, const float* bhn, const float* h0, float* hs, float* hprev,
                      float* gates, int32_t n_seq, int32_t T, int32_t n_inner, void* stream);
/* BPTT over the T steps of every sequence: dgi [rows,192], dgh [rows,192] (then dWh = hprev^T dgh).               */
int32_t dgppo_gru_bwd(const float* dhs, const float* Wh, const float* hprev, const float* gates, float* dgi,
                      float* dgh, int32_t n_seq, int32_t T, int32_t n_inner, void* stream);

/* LSTM scan (train.py --use-lstm; dgppo/nn/rnn.py:22-24 with flax nn.LSTMCell(64): i,f,o sigmoid, g tanh, c' = f c + i g,
 * h' = o tanh(c'), carry (c, h)).  zi [rows,256] = x [W_ii|W_if|W_ig|W_io] precomputed (flax's input Denses have no bias);
 * Wh [64,256], bh [256] (the hidden Denses' biases); row addressing as dgppo_gru_fwd.  c0 / h0 [n_seq,64] or NULL.
 * cs / hs [rows,64]; cprev, hprev [rows,64] and gates [rows,256] (post-activation) are saved for the backward when non-NULL. */
int32_t dgppo_lstm_fwd(const float* zi, const float* Wh, const float* bh, const float* c0, const float* h0, float* cs,
                       float* hs, float* cprev, float* hprev, float* gates, int32_t n_seq, int32_t T, int32_t n_inner,
                       void* stream);
/* BPTT of the above from the gradient of every step's output: dz [rows,256] = gradient of the gate pre-activations
 * (then dW_i = x^T dz, dW_h = hprev^T dz, db_h = colsum dz, dx = dz W_i^T outside).                                     */
int32_t dgppo_lstm_bwd(const float* dhs, const float* Wh, const float* cprev, const float* gates, float* dz,
                       int32_t n_seq, int32_t T, int32_t n_inner, void* stream);

/* tanh-Normal head (algo/module/policy.py:62-74,191-212 ; distribution.py:10-46).  ms [rows,4] = mean(2)|std_trans(2).
 * mode 0 sample (eps [rows,2]) -> action, log_pi; 1 mode -> action = tanh(mean); 2 eval (action_in, eps = the constant
 * entropy noise [n_agents,2]) -> log_pi, entropy and, when dms != NULL, the PPO clipped-surrogate loss gradient
 * (informarl.py:428-438) w.r.t. ms plus stats[0..3] += sum loss, sum entropy, sum(l2>l1), sum|rho-1|.               */
int32_t dgppo_policy_head(const float* ms, const float* eps, const float* action_in, float* action, float* log_pi,
                          float* entropy, int32_t rows, int32_t n_agents, int32_t mode, const float* log_pi_old,
                          const float* adv, float* dms, float* stats, float clip_eps, float coef_ent, void* stream);

/* optax.l2_loss(v, target).mean() (informarl.py:374, dgppo.py:310): dv = (v-target)/count, stats[0] += sum 1/2 d^2 */
int32_t dgppo_value_loss(const float* v, const float* target, float* dv, float* stats, int32_t count, void* stream);
/* mean over the n agents of each graph (value.py:33): x [G,n,D] -> y [G,D]; backward = 1: x = dy [G,D] -> y = dx [G,n,D]
 * (= x / n); backward = 2: y[g,i,:] += x[g,:] (broadcast-add of a per-graph row, the tiled global feature of
 * DecRStateFn(use_global_info=True), value.py:66-68); both optionally through the ReLU that produced the pooled rows
 * (relu_mask [G,n,D] = that ReLU's output, or NULL)                                                                      */
int32_t dgppo_mean_agents(const float* x, float* y, int32_t G, int32_t n, int32_t D, int32_t backward,
                          const float* relu_mask, void* stream);
/* dy *= (y > 0), in place                                                                                             */
int32_t dgppo_relu_bwd(float* dy, const float* y, int64_t count, void* stream);

/* ---- GAE, advantage, optimiser ---------------------------------------------------------------- */

/* compute_dec_ocp_gae (dgppo/algo/utils.py:11-79) for B envs (env-major): costs [B,T,n,nh], rewards [B,T] (l = -reward),
 * Vh [B,T+1,n,nh], Vl [B,T+1], lam_pow [T+1] = lambda^i  ->  Qh [B,T,n,nh], Ql [B,T].                                  */
int32_t dgppo_gae(const float* costs, const float* rewards, const float* Vh, const float* Vl, const float* lam_pow,
                  float gamma, float one_minus_gamma, float one_minus_lam, float* Qh, float* Ql, int32_t B, int32_t T,
                  int32_t n, int32_t nh, void* stream);
/* advantage block (dgppo/algo/dgppo.py:239-259): per-env normalised Ql-Vl, CBF derivative, safe gate, schedule weight
 * -> adv [B,T,n] (already negated); stats[0] += number of safe (t, agent) pairs (eval/safe_data numerator).
 * Vh == NULL: InforMARL's advantage (dgppo/algo/informarl.py:334-336): -(Ql-Vl normalised per env), no CBF terms.    */
int32_t dgppo_advantage(const float* Ql, const float* Vl, const float* Vh, float dt, float alpha, float cbf_eps,
                        float cbf_weight, float* adv, float* stats, int32_t B, int32_t T, int32_t n, int32_t nh,
                        void* stream);
/* InforMARL-Lagrangian advantage (dgppo/algo/informarl_lagr.py:219-235): Al = (Ql - Vl) standardised per env over T, Ah =
 * (Qh - Vh[:, :T]) standardised per (env, agent, component) over T (population std, + 1e-8), adv = -Al - mean_h(Ah *
 * lagr[a,h]).  Ql [B,T], Vl [B,T+1], Qh [B,T,n,nh], Vh [B,T+1,n,nh], lagr [n,nh] -> adv [B,T,n], Ah [B,T,n,nh].          */
int32_t dgppo_advantage_lagr(const float* Ql, const float* Vl, const float* Qh, const float* Vh, const float* lagr,
                             float* adv, float* Ah, int32_t B, int32_t T, int32_t n, int32_t nh, void* stream);
/* update_lagr (informarl_lagr.py:286-309) on one minibatch of n_env envs: delta[a,h] = -mean_{env,t}(Vh (1 - gamma) +
 * exp(lp_new - lp_old) Ah), lagr = relu(lagr - delta * lr).  lp_* [n_env,T,n], Ah [n_env,T,n,nh] (gathered), Vh: base of
 * the gathered value rows with vh_env_stride floats between envs (first T steps used), sums [n*nh]: zeroed scratch.    */
int32_t dgppo_lagr_update(const float* lp_new, const float* lp_old, const float* Vh, int64_t vh_env_stride,
                          const float* Ah, float* lagr, float* sums, int32_t n_env, int32_t T, int32_t n, int32_t nh,
                          float one_minus_gamma, float lr, void* stream);
/* The same step in two halves for the data-parallel update (the `.mean()` of informarl_lagr.py:300-304 runs over the GLOBAL
 * minibatch): dgppo_lagr_sums accumulates this rank's share into sums[n*nh] (+=), the caller all-reduces sums, and
 * dgppo_lagr_apply does lagr = relu(lagr + lr * sums / rows_total) with rows_total = global n_env * T, then zeroes sums. */
int32_t dgppo_lagr_sums(const float* lp_new, const float* lp_old, const float* Vh, int64_t vh_env_stride, const float* Ah,
                        float* sums, int32_t n_env, int32_t T, int32_t n, int32_t nh, float one_minus_gamma, void* stream);
int32_t dgppo_lagr_apply(float* lagr, float* sums, int32_t count, int64_t rows_total, float lr, void* stream);
/* out = max(x, 0): jnp.clip(rollout.costs, a_min=0) of informarl_lagr.py:213                                             */
int32_t dgppo_relu_fwd(const float* x, float* out, int64_t count, void* stream);
/* InforMARL's stage cost l = -reward + w * sum_{agents, components} max(cost, 0) (dgppo/algo/informarl.py:329), as the
 * equivalent reward out = reward - w * sum(...) for dgppo_gae.  reward/out [rows], cost [rows, n, nh].              */
int32_t dgppo_shaped_reward(const float* reward, const float* cost, float cost_weight, float* out, int64_t rows,
                            int32_t n, int32_t nh, void* stream);
/* compute_norm_and_clip + has_any_nan_or_inf (dgppo/trainer/utils.py:89-118) and optax.apply_if_finite(adam)
 * (dgppo/algo/informarl.py:131-137) on one flat buffer.  state [DGPPO_OPT_STATE_FLOATS] lives on the device (zero it once):
 * [2] adam count, [3] total steps, [4] last grad norm, [5] last non-finite flag, [8..] per-workgroup partial sums of the
 * norm / non-finite statistics — reduced in a FIXED order with no atomics, so that data-parallel replicas that start
 * from identical gradients stay bit-identical.
 * grad_scale multiplies every gradient entry as it is read (norm, non-finite test and update all see g * grad_scale):
 * 1 for a single device, 1/world after dgppo_comm_allreduce_sum_f32 (the mean of equal-sized shards' gradients is the
 * gradient of the global mean loss), so the data-parallel path needs no separate scaling pass.                       */
#define DGPPO_OPT_PARTIALS 256
#define DGPPO_OPT_STATE_FLOATS (8 + 2 * DGPPO_OPT_PARTIALS)
int32_t dgppo_clip_adam_step(float* params, const float* grads, float* m, float* v, int64_t n, float* state, float lr,
                             float b1, float b2, float eps, float max_norm, float grad_scale, void* stream);

/* ---- C1: gradient exchange over RCCL / xGMI (no reference counterpart: the reference is single-device, SURVEY F2;
 * this is jax.lax.pmean of the gradient trees of a pmap'ed update_inner, dgppo/algo/dgppo.py:188-294) -------------- */
#define DGPPO_COMM_ID_BYTES 128
/* rank 0: a fresh rendezvous id (ncclGetUniqueId); the caller ships the 128 bytes to the other ranks out of band.   */
int32_t dgppo_comm_unique_id(uint8_t* id_out);
/* version code of the RCCL library the process resolved (ncclGetVersion), for the diagnostics a multi-GPU launch prints
 * before its first collective; no device work.                                                                       */
int32_t dgppo_comm_version(int32_t* version_out);
/* every rank, after selecting its HIP device: join the communicator (ncclCommInitRank).  *comm_out is an opaque
 * handle owned by the library until dgppo_comm_destroy.  Collective: blocks until all `world` ranks have called it. */
int32_t dgppo_comm_init(const uint8_t* id, int32_t rank, int32_t world, void** comm_out);
/* in-place all-reduce(sum) of buf[count] fp32 (DEVICE pointer) enqueued on `stream`; never synchronises the host.
 * One call per minibatch on the flat [g_policy | g_Vl | g_Vh | scalars] buffer.                        */
int32_t dgppo_comm_allreduce_sum_f32(void* comm, float* buf, int64_t count, void* stream);
int32_t dgppo_comm_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif /* DGPPO_HIP_H */
