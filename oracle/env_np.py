"""
ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product path (dgppo_amd/).

NumPy-fp32 CPU restatement of the reference's environment hot path, batched over a leading
axis B.  Every function cites the reference file:line (relative to /root/reference) it follows.

PARITY UNPINNED: the reference (pure JAX) cannot be imported in the build container (jax, flax,
optax, jraph, tensorflow_probability are absent and there is no network) and it ships no tests or
golden vectors, so this restatement is pinned only by the analytic known-answer tests in
tests/test_oracle_env.py, not by outputs of the reference itself.

All arithmetic is IEEE fp32 with an explicit, sequential operation order (no np.sum / np.linalg.norm
on the small axes) so that the HIP kernels — compiled with -ffp-contract=off — can match it bit for bit.
"""
from __future__ import annotations

import dataclasses
import numpy as np

f32 = np.float32

LIDAR_SPREAD, LIDAR_TARGET, LIDAR_BICYCLE_TARGET, MPE_SPREAD, MPE_TARGET = range(5)
# task variants: lidar_env/lidar_line.py, mpe/mpe_line.py, mpe_formation.py, mpe_corridor.py, mpe_connect_spread.py
LIDAR_LINE, MPE_LINE, MPE_FORMATION, MPE_CORRIDOR, MPE_CONNECT_SPREAD = range(5, 10)
KIND_NAMES = {
    "LidarSpread": LIDAR_SPREAD, "LidarTarget": LIDAR_TARGET, "LidarBicycleTarget": LIDAR_BICYCLE_TARGET,
    "MPESpread": MPE_SPREAD, "MPETarget": MPE_TARGET, "LidarLine": LIDAR_LINE, "MPELine": MPE_LINE,
    "MPEFormation": MPE_FORMATION, "MPECorridor": MPE_CORRIDOR, "MPEConnectSpread": MPE_CONNECT_SPREAD,
}
# how the n reward goals follow from the goal nodes (landmarks): the nodes themselves; n points on the segment between two
# landmarks, ends included (lidar_line.py:131-136, mpe_line.py:124-133 for n > 3); n interior points (mpe_line.py n <= 3);
# n points on a circle of radius comm_radius around one landmark (mpe_formation.py:94-98)
GOALS_NODES, GOALS_LINE, GOALS_LINE_INTERIOR, GOALS_CIRCLE = range(4)
RECT_STRIDE = 16


@dataclasses.dataclass
class EnvCfg:
    """PARAMS of dgppo/env/lidar_env/lidar_spread.py:13-22, dgppo/env/mpe/mpe_spread.py:12-19 and the
    make_env defaults of dgppo/env/__init__.py:29-53."""
    kind: int
    n_agents: int
    n_obs: int = 3
    n_rays: int = 32
    top_k: int = 8
    area_size: float = 1.5
    dt: float = 0.03
    car_radius: float = 0.05
    comm_radius: float = 0.5
    obs_radius: float = 0.05
    dist2goal: float = 0.01

    connect_radius: float = 0.45          # MPEConnectSpread (mpe_connect_spread.py:24)
    corridor_width: float = 0.2           # MPECorridor (mpe_corridor.py:19)

    def __post_init__(self):
        # PARAMS the variants override (make_env passes area_size=None: the class default applies)
        if self.kind in (MPE_CORRIDOR, MPE_CONNECT_SPREAD) and self.area_size == 1.5:
            self.area_size = 1.0                                     # default_area_size of both
        if self.kind == MPE_CORRIDOR:
            self.n_obs = 2                                           # mpe_corridor.py:35-37
            self.obs_radius = (self.area_size - self.corridor_width) / 4   # :39
        if self.kind == MPE_CONNECT_SPREAD:
            self.n_obs = 1                                           # mpe_connect_spread.py:38-40
            if self.obs_radius == 0.05:
                self.obs_radius = 0.25

    @property
    def is_lidar(self):
        return self.kind in (LIDAR_SPREAD, LIDAR_TARGET, LIDAR_BICYCLE_TARGET, LIDAR_LINE)

    @property
    def is_spread(self):
        """every agent is connected to every goal node (lidar_spread.py:70-76, mpe_spread.py:63-69 and their subclasses)"""
        return self.kind not in (LIDAR_TARGET, LIDAR_BICYCLE_TARGET, MPE_TARGET)

    @property
    def is_bicycle(self):
        return self.kind == LIDAR_BICYCLE_TARGET

    @property
    def n_goals(self):
        if self.kind in (LIDAR_LINE, MPE_LINE):
            return 2
        if self.kind == MPE_FORMATION:
            return 1
        return self.n_agents

    @property
    def reward_goals(self):
        if self.kind == LIDAR_LINE:
            return GOALS_LINE
        if self.kind == MPE_LINE:
            return GOALS_LINE if self.n_agents > 3 else GOALS_LINE_INTERIOR
        if self.kind == MPE_FORMATION:
            return GOALS_CIRCLE
        return GOALS_NODES

    @property
    def n_cost(self):
        return 3 if self.kind == MPE_CONNECT_SPREAD else 2

    @property
    def obs_mask_radius(self):
        """MPE agent-obstacle edges: within comm_radius, or always connected (mpe_corridor.py:93, mpe_connect_spread.py:169)"""
        return self.comm_radius * 100 if self.kind in (MPE_CORRIDOR, MPE_CONNECT_SPREAD) else self.comm_radius

    @property
    def y_limit(self):
        """upper state limit in y (mpe_corridor.py:62-65, mpe_connect_spread.py:140-143)"""
        return self.area_size * 2 if self.kind in (MPE_CORRIDOR, MPE_CONNECT_SPREAD) else self.area_size

    @property
    def state_dim(self):
        return 5 if self.is_bicycle else 4

    @property
    def node_dim(self):
        return self.state_dim + 3

    @property
    def n_hits(self):
        # lidar_env/base.py:228
        return self.top_k * self.n_agents if (self.is_lidar and self.n_obs > 0) else 0

    @property
    def n_obs_nodes(self):
        return self.n_hits if self.is_lidar else self.n_obs

    @property
    def num_nodes(self):  # incl. pad node, graph.py:231
        return self.n_agents + self.n_goals + self.n_obs_nodes + 1

    @property
    def n_goal_edges(self):
        n = self.n_agents
        return n * self.n_goals if self.is_spread else n

    @property
    def num_edges(self):
        n = self.n_agents
        per_agent_obs = (self.top_k if self.n_obs > 0 else 0) if self.is_lidar else self.n_obs
        return n * n + self.n_goal_edges + n * per_agent_obs

    @property
    def vel_limit(self):
        return 0.5 if self.is_lidar else 1.0


# --------------------------------------------------------------------------------------------------
# helpers with explicit fp32 op order
# --------------------------------------------------------------------------------------------------
def seq_sum(x, axis):
    """left-to-right fp32 sum along a small axis."""
    x = np.moveaxis(x, axis, 0)
    acc = x[0].astype(f32)
    for j in range(1, x.shape[0]):
        acc = (acc + x[j]).astype(f32)
    return acc


def nan_min(x, axis):
    """NaN-propagating min (jnp.min semantics), left to right."""
    x = np.moveaxis(x, axis, 0)
    acc = x[0]
    for j in range(1, x.shape[0]):
        b = x[j]
        acc = np.where(np.isnan(acc) | np.isnan(b), f32(np.nan), np.minimum(acc, b)).astype(f32)
    return acc


def norm2(dx, dy):
    return np.sqrt((dx * dx + dy * dy).astype(f32)).astype(f32)


def ray_table(n_rays):
    """env/utils.py:51  thetas = linspace(-pi, pi - 2pi/R, R); returned as fp32 cos/sin tables."""
    thetas = np.linspace(-np.pi, np.pi - 2 * np.pi / n_rays, n_rays).astype(f32)
    return np.cos(thetas).astype(f32), np.sin(thetas).astype(f32)


def make_rect(center, width, height, theta):
    """Rectangle.create (env/obstacle.py:39-56) -> [..., 16] records (layout: include/dgppo_hip.h)."""
    center = np.asarray(center, f32)
    width = np.asarray(width, f32)
    height = np.asarray(height, f32)
    theta = np.asarray(theta, f32)
    c = np.cos(theta).astype(f32)
    s = np.sin(theta).astype(f32)
    return rect_from_trig(center, width, height, theta, c, s)


def rect_from_trig(center, width, height, theta, c, s):
    hw = (width / f32(2)).astype(f32)
    hh = (height / f32(2)).astype(f32)
    bx = [hw, -hw, -hw, hw]
    by = [hh, hh, -hh, -hh]
    rec = np.zeros(center.shape[:-1] + (RECT_STRIDE,), f32)
    rec[..., 0:2] = center
    rec[..., 2] = width
    rec[..., 3] = height
    rec[..., 4] = theta
    rec[..., 5] = c
    rec[..., 6] = s
    for m in range(4):
        # points = rot @ bbox + center : x = c*bx + (-s)*by + cx ; y = s*bx + c*by + cy
        rec[..., 8 + 2 * m] = ((c * bx[m]).astype(f32) + ((-s) * by[m]).astype(f32)).astype(f32) + center[..., 0]
        rec[..., 9 + 2 * m] = ((s * bx[m]).astype(f32) + (c * by[m]).astype(f32)).astype(f32) + center[..., 1]
    return rec


def rect_inside(px, py, rec, r):
    """Rectangle.inside (env/obstacle.py:62-72).  px,py broadcast against rec[..., 16]."""
    r = f32(r)
    rel_x = (px - rec[..., 0]).astype(f32)
    rel_y = (py - rec[..., 1]).astype(f32)
    c, s = rec[..., 5], rec[..., 6]
    rel_xx = (np.abs((rel_x * c).astype(f32) + (rel_y * s).astype(f32)).astype(f32) - (rec[..., 2] / f32(2)).astype(f32)).astype(f32)
    rel_yy = (np.abs((rel_x * s).astype(f32) - (rel_y * c).astype(f32)).astype(f32) - (rec[..., 3] / f32(2)).astype(f32)).astype(f32)
    is_in_down = (rel_xx < r) & (rel_yy < 0)
    is_in_up = (rel_xx < 0) & (rel_yy < r)
    is_out_corner = (rel_xx > 0) & (rel_yy > 0)
    is_in_circle = np.sqrt(((rel_xx * rel_xx).astype(f32) + (rel_yy * rel_yy).astype(f32)).astype(f32)) < r
    return is_in_down | is_in_up | (is_out_corner & is_in_circle)


# --------------------------------------------------------------------------------------------------
# dynamics  (lidar_env/base.py:142-149, mpe/base.py:129-135, lidar_bicycle_target.py:92-123, env/base.py:80-86)
# --------------------------------------------------------------------------------------------------
def clip_action(action):
    return np.minimum(np.maximum(action.astype(f32), f32(-1.0)), f32(1.0)).astype(f32)


def state_limits(cfg: EnvCfg):
    a = f32(cfg.area_size)
    if cfg.is_bicycle:
        lo = np.array([0, 0, -1, -1, -0.5], f32)
        hi = np.array([a, a, 1, 1, 0.5], f32)
    else:
        v = f32(cfg.vel_limit)
        lo = np.array([0, 0, -v, -v], f32)
        hi = np.array([a, f32(cfg.y_limit), v, v], f32)
    return lo, hi


def agent_step_euler(cfg: EnvCfg, agent, action):
    """agent [B,n,sd], action [B,n,2] (already clipped)."""
    dt = f32(cfg.dt)
    with np.errstate(all="ignore"):
        if cfg.is_bicycle:
            x = agent
            theta = np.arctan2(x[..., 3], x[..., 2]).astype(f32)
            # theta + x[4]*u[0]*dt*10
            theta_next = (theta + (((x[..., 4] * action[..., 0]).astype(f32) * dt).astype(f32) * f32(10)).astype(f32)).astype(f32)
            nx = np.stack([
                (x[..., 0] + ((x[..., 4] * np.cos(theta).astype(f32)).astype(f32) * dt).astype(f32)).astype(f32),
                (x[..., 1] + ((x[..., 4] * np.sin(theta).astype(f32)).astype(f32) * dt).astype(f32)).astype(f32),
                np.cos(theta_next).astype(f32),
                np.sin(theta_next).astype(f32),
                (x[..., 4] + ((action[..., 1] * dt).astype(f32) * f32(10.0)).astype(f32)).astype(f32),
            ], axis=-1)
        else:
            x_dot = np.concatenate([agent[..., 2:], (action * f32(10.0)).astype(f32)], axis=-1)
            nx = ((x_dot * dt).astype(f32) + agent).astype(f32)
    lo, hi = state_limits(cfg)
    return np.minimum(np.maximum(nx, lo), hi).astype(f32)


def state2feat(cfg: EnvCfg, state):
    """lidar_bicycle_target.py:113-118 (identity elsewhere)."""
    if cfg.is_bicycle:
        vx = (state[..., 4] * state[..., 2]).astype(f32)
        vy = (state[..., 4] * state[..., 3]).astype(f32)
        return np.stack([state[..., 0], state[..., 1], vx, vy], axis=-1).astype(f32)
    return state[..., :4].astype(f32)


# --------------------------------------------------------------------------------------------------
# LiDAR  (env/utils.py:49-55,115-136 ; obstacle.py:62-105)
# --------------------------------------------------------------------------------------------------
def lidar_alphas(cfg: EnvCfg, pos, obst, ray_cos, ray_sin):
    """pos [B,n,2], obst [B,n_obs,16] -> alphas [B,n,R], ends [B,n,R,2]."""
    sr = f32(cfg.comm_radius)
    x1 = pos[..., 0][..., None]
    y1 = pos[..., 1][..., None]                                # [B,n,1]
    x2 = (x1 + (ray_cos * sr).astype(f32)).astype(f32)         # [B,n,R]
    y2 = (y1 + (ray_sin * sr).astype(f32)).astype(f32)
    B, n, R = x2.shape
    alphas_o = []
    with np.errstate(all="ignore"):
        for o in range(cfg.n_obs):
            rec = obst[:, o][:, None, None, :]                 # [B,1,1,16]
            per_seg = []
            for m in range(4):
                x3 = rec[..., 8 + 2 * m]
                y3 = rec[..., 9 + 2 * m]
                mm = (m - 1) % 4
                x4 = rec[..., 8 + 2 * mm]
                y4 = rec[..., 9 + 2 * mm]
                x1b, y1b = x1, y1
                det = (((x1b - x2).astype(f32) * (y4 - y3).astype(f32)).astype(f32)
                       - ((y1b - y2).astype(f32) * (x4 - x3).astype(f32)).astype(f32)).astype(f32)
                det = (np.sign(det).astype(f32) * np.minimum(np.maximum(np.abs(det), f32(1e-7)), f32(1e7)).astype(f32)).astype(f32)
                al = ((((y4 - y3).astype(f32) * (x1b - x3).astype(f32)).astype(f32)
                       - ((x4 - x3).astype(f32) * (y1b - y3).astype(f32)).astype(f32)).astype(f32) / det).astype(f32)
                be = ((((-(y1b - y2).astype(f32)) * (x1b - x3).astype(f32)).astype(f32)
                       + ((x1b - x2).astype(f32) * (y1b - y3).astype(f32)).astype(f32)).astype(f32) / det).astype(f32)
                valid = ((al <= 1) & (al >= 0) & (be <= 1) & (be >= 0)).astype(f32)
                al = ((valid * al).astype(f32) + ((f32(1) - valid) * f32(1e6)).astype(f32)).astype(f32)
                per_seg.append(al)
            alphas_o.append(nan_min(np.stack(per_seg, 0), 0))
        alphas = nan_min(np.stack(alphas_o, 0), 0)             # [B,n,R]
        # is_in (r = 0), env/utils.py:117,129
        is_in = np.zeros((B, n), bool)
        for o in range(cfg.n_obs):
            is_in |= rect_inside(pos[..., 0], pos[..., 1], obst[:, o][:, None, :], 0.0)
        alphas = (alphas * (f32(1) - is_in.astype(f32))[..., None]).astype(f32)
    ends = np.stack([np.broadcast_to(x2, (B, n, R)), np.broadcast_to(y2, (B, n, R))], axis=-1)
    return alphas, ends


def lidar_sense(cfg: EnvCfg, pos, obst, ray_cos, ray_sin):
    """get_lidar_data (lidar_env/base.py:126-140): -> hits [B,n,k,2] (sorted by stable ascending alpha)."""
    alphas, ends = lidar_alphas(cfg, pos, obst, ray_cos, ray_sin)
    idx = np.argsort(alphas, axis=-1, kind="stable")[..., :cfg.top_k]        # NaN last, ties keep ray order
    start = pos[:, :, None, :]
    with np.errstate(all="ignore"):
        hit = (start + ((ends - start).astype(f32) * alphas[..., None]).astype(f32)).astype(f32)  # [B,n,R,2]
    return np.take_along_axis(hit, idx[..., None], axis=2).astype(f32), idx


# --------------------------------------------------------------------------------------------------
# reward / cost   (lidar_spread.py:35-52, lidar_target.py:35-52, mpe twins ; lidar_env/base.py:180-207, mpe/base.py:164-191)
# --------------------------------------------------------------------------------------------------
def reward_goal_positions(cfg: EnvCfg, goal):
    """[B, n_goals, >=2] goal nodes -> [B, n_reward_goals, 2] positions the reward measures against."""
    gp = goal[..., :2].astype(f32)
    mode, n = cfg.reward_goals, cfg.n_agents
    if mode == GOALS_NODES:
        return gp
    if mode in (GOALS_LINE, GOALS_LINE_INTERIOR):
        # landmark2goal: goals = l0 + arange(...)[:, None] * (l1 - l0) / n_interval, evaluated left to right in fp32
        direction = (gp[:, 1] - gp[:, 0]).astype(f32)                                    # [B,2]
        if mode == GOALS_LINE:
            idx, n_interval = np.arange(0, n, dtype=f32), f32(n - 1)
        else:
            idx, n_interval = np.arange(1, n + 1, dtype=f32), f32(n + 1)
        step = ((idx[None, :, None] * direction[:, None, :]).astype(f32) / n_interval).astype(f32)
        return (gp[:, 0][:, None, :] + step).astype(f32)
    # GOALS_CIRCLE: thetas = linspace(0, 2 pi, n + 1)[:-1]  [upstream: jnp.linspace = start + (iota / div) * delta],
    # goals = landmark + R * [cos, sin]
    t = (np.arange(n, dtype=f32) / f32(n)).astype(f32)
    th = (t * f32(2 * np.pi)).astype(f32)
    R = f32(cfg.comm_radius)
    off = np.stack([(R * np.cos(th).astype(f32)).astype(f32), (R * np.sin(th).astype(f32)).astype(f32)], axis=-1)
    return (gp[:, 0][:, None, :] + off[None]).astype(f32)


def get_reward(cfg: EnvCfg, agent, goal, action):
    ap = agent[..., :2]
    gp = reward_goal_positions(cfg, goal)
    n = cfg.n_agents
    if cfg.is_spread:
        d = norm2(gp[:, :, None, 0] - ap[:, None, :, 0], gp[:, :, None, 1] - ap[:, None, :, 1])   # [B,g,j]
        dist2goal = nan_min(d, 2)
    else:
        dist2goal = norm2(gp[..., 0] - ap[..., 0], gp[..., 1] - ap[..., 1])
    ng = dist2goal.shape[1]
    reward = np.zeros(agent.shape[0], f32)
    reward = (reward - ((seq_sum(dist2goal, 1) / f32(ng)).astype(f32) * f32(0.01)).astype(f32)).astype(f32)
    ind = np.where(dist2goal > f32(cfg.dist2goal), f32(1.0), f32(0.0)).astype(f32)
    reward = (reward - ((seq_sum(ind, 1) / f32(ng)).astype(f32) * f32(0.001)).astype(f32)).astype(f32)
    an = norm2(action[..., 0], action[..., 1])
    an2 = (an * an).astype(f32)
    reward = (reward - ((seq_sum(an2, 1) / f32(n)).astype(f32) * f32(0.0001)).astype(f32)).astype(f32)
    return reward


def get_cost(cfg: EnvCfg, agent, hits_or_obs):
    ap = agent[..., :2]
    B, n = ap.shape[:2]
    d = norm2(ap[:, :, None, 0] - ap[:, None, :, 0], ap[:, :, None, 1] - ap[:, None, :, 1])
    d = (d + (np.eye(n, dtype=f32) * f32(1e6)).astype(f32)).astype(f32)
    min_dist = nan_min(d, 2)
    agent_cost = (f32(cfg.car_radius * 2) - min_dist).astype(f32)
    if cfg.n_obs == 0:
        obs_cost = np.zeros((B, n), f32)
    elif cfg.is_lidar:
        hp = hits_or_obs                                                       # [B,n,k,2]
        dd = norm2(hp[..., 0] - ap[:, :, None, 0], hp[..., 1] - ap[:, :, None, 1])
        obs_cost = (f32(cfg.car_radius) - nan_min(dd, 2)).astype(f32)
    else:
        op = hits_or_obs[..., :2]                                              # [B,n_obs,2]
        dd = norm2(ap[:, :, None, 0] - op[:, None, :, 0], ap[:, :, None, 1] - op[:, None, :, 1])
        obs_cost = (f32(cfg.car_radius + cfg.obs_radius) - nan_min(dd, 2)).astype(f32)
    comps = [agent_cost, obs_cost]
    if cfg.n_cost == 3:
        # connectivity (mpe_connect_spread.py:115-117): the largest nearest-neighbour distance against connect_radius, the
        # same value for every agent
        with np.errstate(invalid="ignore"):
            worst = (min_dist - f32(cfg.connect_radius)).astype(f32)
        cc = worst[:, 0].copy()
        for j in range(1, n):                                        # jnp.max: NaN propagates
            cc = np.where(np.isnan(cc) | np.isnan(worst[:, j]), f32(np.nan), np.maximum(cc, worst[:, j])).astype(f32)
        comps.append(np.broadcast_to(cc[:, None], (B, n)).astype(f32))
    cost = np.stack(comps, axis=-1)
    cost = np.where(cost <= 0.0, (cost - f32(0.5)).astype(f32), (cost + f32(0.5)).astype(f32)).astype(f32)
    if cfg.is_lidar or cfg.kind == MPE_CONNECT_SPREAD:               # mpe_connect_spread.py:134 clips both sides
        cost = np.minimum(np.maximum(cost, f32(-1.0)), f32(1.0))
    else:
        cost = np.maximum(cost, f32(-1.0))            # mpe/base.py:189 clips only from below
    return cost.astype(f32)


# --------------------------------------------------------------------------------------------------
# graph   (lidar_env/base.py:227-271, mpe/base.py:211-241, lidar_spread.py:57-96, lidar_target.py:57-96,
#          mpe_spread.py:51-81, mpe_target.py:51-80, utils/graph.py:35-44,212-247)
# --------------------------------------------------------------------------------------------------
def _edge_block(feats, mask, ids_recv, ids_send, pad_id):
    """EdgeBlock.make_edges (graph.py:35-44); feats [B,r,s,4], mask [B,r,s]."""
    B, r, s = mask.shape
    recv = np.where(mask, np.broadcast_to(ids_recv[None, :, None], (B, r, s)), pad_id).astype(np.int32)
    send = np.where(mask, np.broadcast_to(ids_send[None, None, :], (B, r, s)), pad_id).astype(np.int32)
    return feats.reshape(B, r * s, -1).astype(f32), recv.reshape(B, r * s), send.reshape(B, r * s)


def get_graph(cfg: EnvCfg, agent, goal, obst, hits):
    """-> dict(nodes, edges, states, receivers, senders, node_type, n_node, n_edge), all batched [B,...].
    LiDAR: obst = rectangle records (unused for the graph), hits [B,n,k,2].  MPE: obst [B,n_obs,sd]."""
    B = agent.shape[0]
    n, ng, sd, nd = cfg.n_agents, cfg.n_goals, cfg.state_dim, cfg.node_dim
    N = cfg.num_nodes
    pad_id = N - 1
    n_on = cfg.n_obs_nodes
    nodes = np.zeros((B, N, nd), f32)
    states = np.zeros((B, N, sd), f32)
    node_type = -np.ones((B, N), np.int32)
    nodes[:, :n, :sd] = agent
    nodes[:, n:n + ng, :sd] = goal
    nodes[:, :n, sd + 2] = 1.0
    nodes[:, n:n + ng, sd + 1] = 1.0
    states[:, :n] = agent
    states[:, n:n + ng] = goal
    node_type[:, :n] = 0
    node_type[:, n:n + ng] = 1
    if n_on > 0:
        if cfg.is_lidar:
            flat = hits.reshape(B, n_on, 2)
            nodes[:, n + ng:n + ng + n_on, :2] = flat
            states[:, n + ng:n + ng + n_on, :2] = flat
        else:
            nodes[:, n + ng:n + ng + n_on, :sd] = obst
            states[:, n + ng:n + ng + n_on] = obst
        nodes[:, n + ng:n + ng + n_on, sd] = 1.0
        node_type[:, n + ng:n + ng + n_on] = 2
    states[:, pad_id] = -1.0                                   # graph.py:217-218

    feat_a = state2feat(cfg, agent)
    feat_g = state2feat(cfg, goal)
    ap = agent[..., :2]
    id_agent = np.arange(n)
    blocks = []
    # agent-agent
    d = norm2(ap[:, :, None, 0] - ap[:, None, :, 0], ap[:, :, None, 1] - ap[:, None, :, 1])
    d = (d + (np.eye(n, dtype=f32) * f32(cfg.comm_radius + 1)).astype(f32)).astype(f32)
    aa_mask = d < f32(cfg.comm_radius)
    aa_feats = (feat_a[:, :, None, :] - feat_a[:, None, :, :]).astype(f32)
    blocks.append(_edge_block(aa_feats, aa_mask, id_agent, id_agent, pad_id))
    # agent-goal
    if cfg.is_spread:
        ag_feats = (feat_a[:, :, None, :] - feat_g[:, None, :, :]).astype(f32)
        blocks.append(_edge_block(ag_feats, np.ones((B, n, ng), bool), id_agent, np.arange(n, n + ng), pad_id))
    else:
        for i in range(n):
            f = (feat_a[:, i] - feat_g[:, i]).astype(f32)[:, None, None, :]
            blocks.append(_edge_block(f, np.ones((B, 1, 1), bool), np.array([i]), np.array([i + n]), pad_id))
    # agent-obs
    if cfg.is_lidar:
        if n_on > 0:
            k = cfg.top_k
            for i in range(n):
                lf = (ap[:, i, None, :] - hits[:, i]).astype(f32)                      # [B,k,2]
                ld = norm2(lf[..., 0], lf[..., 1])
                active = ld < f32(cfg.comm_radius - 1e-1)
                lf4 = np.concatenate([lf, np.zeros((B, k, 2), f32)], axis=-1)
                ids = n + ng + i * k + np.arange(k)
                blocks.append(_edge_block(lf4[:, None], active[:, None], np.array([i]), ids, pad_id))
    elif cfg.n_obs > 0:       # MPETarget with n_obs == 0 is guarded like MPESpread (mpe_spread.py:71-72)
        op = obst[..., :2]
        dd = norm2(ap[:, :, None, 0] - op[:, None, :, 0], ap[:, :, None, 1] - op[:, None, :, 1])
        ao_mask = dd < f32(cfg.obs_mask_radius)
        ao_feats = (agent[:, :, None, :4] - obst[:, None, :, :4]).astype(f32)
        blocks.append(_edge_block(ao_feats, ao_mask, id_agent, np.arange(cfg.n_obs) + n + ng, pad_id))
    edges = np.concatenate([b[0] for b in blocks], axis=1)
    recv = np.concatenate([b[1] for b in blocks], axis=1)
    send = np.concatenate([b[2] for b in blocks], axis=1)
    assert edges.shape[1] == cfg.num_edges, (edges.shape, cfg.num_edges)
    return dict(nodes=nodes, edges=edges, states=states, receivers=recv, senders=send, node_type=node_type,
                n_node=np.full((B,), N, np.int32), n_edge=np.full((B,), cfg.num_edges, np.int32))


# --------------------------------------------------------------------------------------------------
# step   (lidar_env/base.py:151-174, mpe/base.py:137-162)
# --------------------------------------------------------------------------------------------------
def env_step(cfg: EnvCfg, agent, goal, obst, hits, action, ray_tab=None, want_graph=True):
    """-> dict(next_agent, next_hits, reward, cost, graph).  `hits` are the hit points of the graph at t."""
    a = clip_action(action)
    next_agent = agent_step_euler(cfg, agent, a)
    next_hits = None
    if cfg.is_lidar and cfg.n_obs > 0:
        rc, rs = ray_tab if ray_tab is not None else ray_table(cfg.n_rays)
        next_hits, _ = lidar_sense(cfg, next_agent[..., :2], obst, rc, rs)
    reward = get_reward(cfg, agent, goal, a)
    cost = get_cost(cfg, agent, hits if cfg.is_lidar else obst)
    graph = get_graph(cfg, next_agent, goal, obst, next_hits) if want_graph else None
    return dict(next_agent=next_agent, next_hits=next_hits, reward=reward, cost=cost, graph=graph)


# --------------------------------------------------------------------------------------------------
# reset   (lidar_env/base.py:89-124, lidar_bicycle_target.py:60-90, mpe/base.py:81-127, env/utils.py:139-244)
# Philox-4x32-10 counter RNG (the JAX threefry streams cannot be reproduced: SURVEY A.4, A.12).
# Stream layout (shared with dgppo_amd/csrc/env_reset.hip): draw d of env with seed s uses
# Philox(counter=(d, 0, 0, 0), key=(lo32(s), hi32(s))) and takes words 0,1 -> two uniforms in [0,1).
# --------------------------------------------------------------------------------------------------
_PH_M0, _PH_M1 = 0xD2511F53, 0xCD9E8D57
_PH_W0, _PH_W1 = 0x9E3779B9, 0xBB67AE85


def philox4x32(counter, key):
    c = [int(x) & 0xFFFFFFFF for x in counter]
    k = [int(x) & 0xFFFFFFFF for x in key]
    for _ in range(10):
        p0 = _PH_M0 * c[0]
        p1 = _PH_M1 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k[0]) & 0xFFFFFFFF, p1 & 0xFFFFFFFF,
             ((p0 >> 32) ^ c[3] ^ k[1]) & 0xFFFFFFFF, p0 & 0xFFFFFFFF]
        k = [(k[0] + _PH_W0) & 0xFFFFFFFF, (k[1] + _PH_W1) & 0xFFFFFFFF]
    return c


def u01(word):
    return f32(f32(word >> 8) * f32(1.0 / 16777216.0))


class _Stream:
    def __init__(self, seed):
        self.key = (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
        self.d = 0

    def uniform2(self):
        w = philox4x32((self.d, 0, 0, 0), self.key)
        self.d += 1
        return u01(w[0]), u01(w[1])


def reset_thresholds(cfg: EnvCfg):
    """thresholds of the variant resets, formed in Python doubles and rounded once to fp32 (as the reference's weakly typed
    Python floats are); the same numbers travel to the device in dgppo_env_cfg."""
    cr, A = cfg.car_radius, cfg.area_size
    d = dict(min_dist=2 * cr, side_y=A, goal_shift_y=0.0, line_min_dist=0.0)
    if cfg.kind in (MPE_CORRIDOR, MPE_CONNECT_SPREAD):
        d["side_y"] = (A - cfg.obs_radius * 2) / 2 - 1.5 * cr                    # mpe_corridor.py:50, mpe_connect_spread.py:79
        d["goal_shift_y"] = A - (A - cfg.obs_radius * 2) / 2 + 1.5 * cr          # :52, :81-83
    if cfg.kind == MPE_CONNECT_SPREAD:
        d["min_dist"] = 2.3 * cr                                                 # :77
    if cfg.kind in (LIDAR_LINE, MPE_LINE):
        n = cfg.n_agents
        d["line_min_dist"] = n * 5 * cr if (cfg.kind == MPE_LINE and n <= 3) else (n - 2) * 6 * cr   # mpe_line.py:49-52
    return {k: f32(v) for k, v in d.items()}


def _sample_pairs(st, n, Ax, Ay, min_dist, max_iter=1024):
    """get_node_goal_rng (env/utils.py:139-244) without obstacles: agents and goals alternately, each at least min_dist from
    the rows already placed (zero rows of the work arrays included), <= max_iter tries each, restart on failure."""
    while True:
        states = np.zeros((n, 2), f32)
        goals = np.zeros((n, 2), f32)
        failed = False
        for i in range(n):
            its = []
            for arr in (states, goals):
                it = 0
                while True:
                    u0, u1 = st.uniform2()
                    cand = np.array([u0 * Ax, u1 * Ay], f32)
                    dmin = np.min(norm2(arr[:, 0] - cand[0], arr[:, 1] - cand[1]))
                    if (not (dmin <= min_dist)) or it >= max_iter:
                        break
                    it += 1
                arr[i] = cand
                its.append(it)
            if max(its) >= max_iter:
                failed = True
                break
        if not failed:
            return states, goals


def _nn_dist(p):
    d = norm2(p[:, None, 0] - p[None, :, 0], p[:, None, 1] - p[None, :, 1])
    d = (d + (np.eye(len(p), dtype=f32) * f32(1e6)).astype(f32)).astype(f32)
    return d.min(axis=1)


def _reset_variant(cfg: EnvCfg, seed: int):
    """lidar_line.py:39-129, mpe_line.py:36-122, mpe_formation.py:37-92, mpe_corridor.py:41-60, mpe_connect_spread.py:50-107.
    Stream: the same Philox draws as env_reset_single, consumed in the order of the statements below.  Deviation: the
    landmark's quarter-turn (region * pi / 2) uses exact rotations instead of fp32 cos / sin of k * pi / 2 (4e-8 off)."""
    st = _Stream(int(seed))
    n, sd, A = cfg.n_agents, cfg.state_dim, f32(cfg.area_size)
    th = reset_thresholds(cfg)
    cr = f32(cfg.car_radius)
    obst = None
    if cfg.kind == MPE_CONNECT_SPREAD:
        for attempt in range(4096):                     # the reference loops without bound (n == 1 would never end)
            states, goals = _sample_pairs(st, n, A, th["side_y"], th["min_dist"])
            mda, mdg = _nn_dist(states), _nn_dist(goals)
            bad = bool(np.any(mda > f32(cfg.connect_radius))) or bool(np.any(mda < f32(2 * cfg.car_radius))) \
                or bool(np.any(mdg > f32(cfg.connect_radius)))
            if not bad:
                break
        goals = goals.copy()
        goals[:, 1] = (goals[:, 1] + th["goal_shift_y"]).astype(f32)
        u0, _ = st.uniform2()
        r_o = f32(cfg.obs_radius)
        ox = f32(r_o + f32(u0 * f32(f32(A - r_o) - r_o)))
        obst = np.zeros((1, sd), f32)
        obst[0, :2] = [ox, f32(A / f32(2))]
        goal_nodes = goals
    elif cfg.kind == MPE_CORRIDOR:
        states, goals = _sample_pairs(st, n, A, th["side_y"], th["min_dist"])
        goals = goals.copy()
        goals[:, 1] = (goals[:, 1] + th["goal_shift_y"]).astype(f32)
        r_o = f32(cfg.obs_radius)
        obst = np.zeros((2, sd), f32)
        obst[0, :2] = [r_o, f32(A / f32(2))]
        obst[1, :2] = [f32(A - r_o), f32(A / f32(2))]
        goal_nodes = goals
    else:
        states, _ = _sample_pairs(st, n, A, A, th["min_dist"])
        if cfg.kind == MPE_FORMATION:
            # formed in double from the fp32 PARAMS, rounded once (the host side of dgppo_env_reset does the same)
            d = lambda v: float(f32(v))
            lo = f32(d(cfg.comm_radius) + 2.0 * d(cfg.car_radius))
            hi = f32(d(cfg.area_size) - d(cfg.comm_radius) - 2.0 * d(cfg.car_radius))
            u0, u1 = st.uniform2()
            landmarks = np.array([[f32(lo + f32(u0 * f32(hi - lo))), f32(lo + f32(u1 * f32(hi - lo)))]], f32)
        else:
            md = th["line_min_dist"]
            if cfg.kind == MPE_LINE and n <= 3:
                u0, u1 = st.uniform2()
                l0 = np.array([u0 * A, u1 * A], f32)
            else:
                side = f32(A - md)
                assert side >= 0, "The area size is too small to place the landmarks."
                u0, u1 = st.uniform2()
                cx = f32(f32(u0 * f32(A - side)) - f32(A / f32(2)))
                cy = f32(f32(u1 * side) + f32(f32(A / f32(2)) - side))
                u0, _ = st.uniform2()
                region = min(int(f32(u0 * f32(4.0))), 3)
                rx, ry = [(cx, cy), (f32(-cy), cx), (f32(-cx), f32(-cy)), (cy, f32(-cx))][region]
                l0 = np.array([f32(rx + f32(A / f32(2))), f32(ry + f32(A / f32(2)))], f32)
            while True:
                u0, u1 = st.uniform2()
                l1 = np.array([u0 * A, u1 * A], f32)
                if not (norm2(l1[0] - l0[0], l1[1] - l0[1]) < md):
                    break
            landmarks = np.stack([l0, l1]).astype(f32)
        goal_nodes = landmarks
        rgoals = reward_goal_positions(cfg, landmarks[None])[0]
        if cfg.is_lidar:
            # lidar_line.py:88-118: rectangles that keep 1.1 car radii from every agent and every line goal
            obst = np.zeros((cfg.n_obs, RECT_STRIDE), f32)
            pts = np.concatenate([states, rgoals], 0)
            r_in = f32(float(f32(cfg.car_radius)) * 1.1)
            for o in range(cfg.n_obs):
                while True:
                    u0, u1 = st.uniform2()
                    cx, cy = f32(u0 * A), f32(u1 * A)
                    u0, u1 = st.uniform2()
                    lo, hi = f32(0.1), f32(0.3)
                    w = f32(lo + f32(u0 * f32(hi - lo)))
                    h = f32(lo + f32(u1 * f32(hi - lo)))
                    u0, _ = st.uniform2()
                    rec = make_rect(np.array([cx, cy], f32), w, h, f32(u0 * f32(np.pi)))
                    if not bool(np.any(rect_inside(pts[:, 0], pts[:, 1], rec[None], r_in))):
                        break
                obst[o] = rec
        else:
            # mpe_line.py:88-112 / mpe_formation.py:55-79: discs, rejection vs agents / reward goals / border
            obst = np.zeros((cfg.n_obs, sd), f32)
            lo = f32(cr * f32(3.0))
            hi = f32(A - lo)
            thr_a = f32(cfg.car_radius + cfg.obs_radius)
            thr_g = f32(f32(cfg.car_radius * 2) + f32(cfg.obs_radius))
            for o in range(cfg.n_obs):
                first = True
                while True:
                    u0, u1 = st.uniform2()
                    if first:
                        cand = np.array([u0 * A, u1 * A], f32)
                        first = False
                    else:
                        cand = np.array([lo + u0 * f32(hi - lo), lo + u1 * f32(hi - lo)], f32)
                    da = np.min(norm2(states[:, 0] - cand[0], states[:, 1] - cand[1]))
                    dg = np.min(norm2(rgoals[:, 0] - cand[0], rgoals[:, 1] - cand[1]))
                    bad = (da <= thr_a) or (dg <= thr_g) or bool(np.any(cand < lo)) or bool(np.any(cand > hi))
                    if not bad:
                        break
                obst[o, :2] = cand
    agent = np.zeros((n, sd), f32)
    goal = np.zeros((cfg.n_goals, sd), f32)
    agent[:, :2] = states
    goal[:, :2] = goal_nodes
    return agent, goal, obst


def env_reset_single(cfg: EnvCfg, seed: int):
    """one env; pure-Python loops (small cases only).  Returns agent [n,sd], goal [n_goals,sd], obst."""
    if cfg.kind >= LIDAR_LINE:
        return _reset_variant(cfg, seed)
    st = _Stream(int(seed))
    n, sd = cfg.n_agents, cfg.state_dim
    A = f32(cfg.area_size)
    max_iter = 1024
    if cfg.is_lidar:
        min_dist = f32(2.2 * cfg.car_radius)
        obst = np.zeros((cfg.n_obs, RECT_STRIDE), f32)
        for o in range(cfg.n_obs):
            u0, u1 = st.uniform2()
            cx, cy = f32(u0 * A), f32(u1 * A)
            u0, u1 = st.uniform2()
            lo, hi = f32(0.1), f32(0.3)
            w = f32(lo + f32(u0 * f32(hi - lo)))
            h = f32(lo + f32(u1 * f32(hi - lo)))
            u0, _ = st.uniform2()
            two_pi = f32(2 * np.pi)
            th = f32(u0 * two_pi)
            if cfg.is_bicycle:
                th = f32(th - f32(np.pi))               # U(-pi, pi), lidar_bicycle_target.py:74
            obst[o] = make_rect(np.array([cx, cy], f32), w, h, th)
    else:
        min_dist = f32(2 * cfg.car_radius)
        obst = None

    def inside_any(px, py, r):
        if not cfg.is_lidar or cfg.n_obs == 0:
            return False
        return bool(np.any(rect_inside(px, py, obst, r)))

    half = f32(min_dist / f32(2))
    while True:
        states = np.zeros((n, 2), f32)
        goals = np.zeros((n, 2), f32)
        failed = False
        for i in range(n):
            it = 0
            while True:
                u0, u1 = st.uniform2()
                cand = np.array([u0 * A, u1 * A], f32)
                dmin = np.min(norm2(states[:, 0] - cand[0], states[:, 1] - cand[1]))
                ok = (not (dmin <= min_dist)) and (not inside_any(cand[0], cand[1], half))
                if ok or it >= max_iter:
                    break
                it += 1
            n_iter_agent = it
            states[i] = cand
            it = 0
            while True:
                u0, u1 = st.uniform2()
                cand = np.array([u0 * A, u1 * A], f32)
                dmin = np.min(norm2(goals[:, 0] - cand[0], goals[:, 1] - cand[1]))
                ok = (not (dmin <= min_dist)) and (not inside_any(cand[0], cand[1], half))
                if ok or it >= max_iter:
                    break
                it += 1
            goals[i] = cand
            if n_iter_agent >= max_iter or it >= max_iter:
                failed = True
                break
        if not failed:
            break
    agent = np.zeros((n, sd), f32)
    goal = np.zeros((n, sd), f32)
    agent[:, :2] = states
    goal[:, :2] = goals
    if cfg.is_bicycle:
        for i in range(n):
            u0, _ = st.uniform2()
            th = f32(u0 * f32(2 * np.pi))
            agent[i, 2] = np.cos(th).astype(f32)
            agent[i, 3] = np.sin(th).astype(f32)
    if not cfg.is_lidar:
        # mpe/base.py:93-118: obstacle discs, rejection vs agents / goals / border
        obst = np.zeros((cfg.n_obs, sd), f32)
        lo = f32(f32(cfg.car_radius) * f32(3.0))
        hi = f32(A - lo)
        thr_a = f32(cfg.car_radius + cfg.obs_radius)
        thr_g = f32(f32(cfg.car_radius * 2) + f32(cfg.obs_radius))
        for o in range(cfg.n_obs):
            first = True
            while True:
                u0, u1 = st.uniform2()
                if first:
                    cand = np.array([u0 * A, u1 * A], f32)
                    first = False
                else:
                    cand = np.array([lo + u0 * f32(hi - lo), lo + u1 * f32(hi - lo)], f32)
                da = np.min(norm2(states[:, 0] - cand[0], states[:, 1] - cand[1]))
                dg = np.min(norm2(goals[:, 0] - cand[0], goals[:, 1] - cand[1]))
                bad = (da <= thr_a) or (dg <= thr_g) \
                    or bool(np.any(cand < lo)) or bool(np.any(cand > hi))
                if not bad:
                    break
            obst[o, :2] = cand
    return agent, goal, obst


def env_reset(cfg: EnvCfg, seeds):
    outs = [env_reset_single(cfg, int(s)) for s in seeds]
    agent = np.stack([o[0] for o in outs])
    goal = np.stack([o[1] for o in outs])
    obst = np.stack([o[2] for o in outs]) if outs[0][2] is not None else None
    return agent, goal, obst
