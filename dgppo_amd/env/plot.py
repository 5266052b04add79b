"""Episode rendering for test.py (the reference's dgppo/env/plot.py:206 `render_mpe` and :468
`render_lidar`, called through `env.render_video`, dgppo/env/lidar_env/base.py:209-221, dgppo/env/mpe/base.py).

Host-side only (matplotlib): one episode's GraphsTuple sequence is copied to NumPy once, then every frame moves the
agent circles, rebuilds the edge segments from (senders, receivers) minus the pad node, and rewrites the cost / reward /
unsafe / step texts.  What a frame shows follows the reference: agents in blue with their index, goals in green,
obstacles in dark red (rectangles for the LiDAR family, discs for MPE), communication edges in grey, agent-goal edges in
green, LiDAR hit points as the far end of agent-hit edges.  The container is whatever matplotlib can write here: `.mp4`
needs an ffmpeg binary (the reference's default); without one the same frames go to an animated `.gif` next to the
requested path (Pillow writer) and the path actually written is returned."""
from __future__ import annotations

import pathlib
from typing import NamedTuple, Optional, Sequence, Tuple

import numpy as np

AGENT_COLOR, GOAL_COLOR, OBS_COLOR, COMM_COLOR = "#0068ff", "#2fdd00", "#8a0000", "0.2"


class Episode(NamedTuple):
    """one episode on the host: T frames of a fixed-topology padded graph"""
    states: np.ndarray        # [T, N, sd]   node states, pad node last
    senders: np.ndarray       # [T, E]
    receivers: np.ndarray     # [T, E]
    rewards: np.ndarray       # [T]
    costs: np.ndarray         # [T, n, n_cost]
    rect_points: Optional[np.ndarray]   # [n_obs, 4, 2] LiDAR rectangles (static within an episode) or None
    discs: Optional[np.ndarray]         # [n_obs, 2] MPE obstacle centres or None


def _np(x) -> np.ndarray:
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    return np.asarray(x)


def episode_from_rollout(rollout, index: Optional[int] = None) -> Episode:
    """Slice one episode out of a Rollout.  `rollout.graph` carries [T, ...] (a single episode, as the reference passes
    it) or [B, T, ...] (this build's batched rollouts: pick `index`)."""
    g = rollout.graph
    batched = g.states.ndim == 4
    if batched and index is None:
        if g.states.shape[0] != 1:
            raise ValueError("batched rollout: pass the episode index")
        index = 0
    pick = (lambda a: _np(a[index])) if batched else _np       # slice on the device, copy one episode
    es = g.env_states
    rect = discs = None
    if getattr(es, "obstacle", None) is not None:
        pts = pick(es.obstacle.points)                         # [T, n_obs, 4, 2] or [n_obs, 4, 2]
        rect = pts[0] if pts.ndim == 4 else pts
    elif getattr(es, "obs", None) is not None:
        o = pick(es.obs)
        discs = (o[0] if o.ndim == 3 else o)[:, :2]
    return Episode(pick(g.states), pick(g.senders), pick(g.receivers), pick(rollout.rewards), pick(rollout.costs), rect,
                   discs)


def frame_edges(ep: Episode, t: int, n_agent: int, n_goal: int) -> Tuple[np.ndarray, np.ndarray]:
    """segments [e, 2, 2] of frame t and a flag per segment: sender is a goal node.  Edges touching the pad node (the last
    node) are dropped — masked edges are re-routed pad -> pad by the graph builder (utils/graph.py:212-247)."""
    pad = ep.states.shape[1] - 1
    s, r = ep.senders[t], ep.receivers[t]
    keep = (s != pad) & (r != pad)
    s, r = s[keep], r[keep]
    pos = ep.states[t, :, :2]
    seg = np.stack([pos[s], pos[r]], axis=1)
    finite = np.isfinite(seg).all(axis=(1, 2))                # a LiDAR hit point can be NaN (parallel ray): not drawable
    return seg[finite], ((s >= n_agent) & (s < n_agent + n_goal))[finite]


def unsafe_agents(ep: Episode, t: int, Ta_is_unsafe=None) -> np.ndarray:
    """indices of the agents flagged unsafe at frame t (test.py:103-105: any cost component >= 0)"""
    if Ta_is_unsafe is not None:
        return np.flatnonzero(_np(Ta_is_unsafe)[t])
    return np.flatnonzero((ep.costs[t] >= 0.0).any(axis=-1))


class _Scene:
    """the artists of one figure; `draw(t)` moves them to frame t"""

    def __init__(self, ep: Episode, side_length: float, n_agent: int, n_goal: int, r: float, obs_r: float,
                 cost_components: Sequence[str], Ta_is_unsafe, dpi: int):
        import matplotlib
        matplotlib.use("Agg", force=False)
        import matplotlib.pyplot as plt
        from matplotlib.collections import LineCollection, PatchCollection
        from matplotlib.patches import Circle, Polygon

        self.ep, self.n_agent, self.n_goal = ep, n_agent, n_goal
        self.cost_components, self.Ta_is_unsafe = tuple(cost_components), Ta_is_unsafe
        self.fig, ax = plt.subplots(1, 1, figsize=(10, 10), dpi=dpi)
        self.ax = ax
        ax.set_xlim(0.0, side_length)
        ax.set_ylim(0.0, side_length)
        ax.set_aspect("equal")
        ax.axis("off")
        ax.add_patch(plt.Rectangle((0, 0), side_length, side_length, fill=False, edgecolor="0.6", linewidth=1.0, zorder=0))
        if ep.rect_points is not None and len(ep.rect_points):
            ax.add_collection(PatchCollection([Polygon(p, closed=True) for p in ep.rect_points], facecolor=OBS_COLOR,
                                              edgecolor="none", alpha=0.8, zorder=1))
        if ep.discs is not None and len(ep.discs):
            ax.add_collection(PatchCollection([Circle(c, obs_r) for c in ep.discs], facecolor=OBS_COLOR, edgecolor="none",
                                              zorder=1))
        pos0 = ep.states[0, :, :2]
        self.goal_circles = [Circle(pos0[n_agent + j], r, color=GOAL_COLOR, linewidth=0.0, zorder=5) for j in range(n_goal)]
        self.agent_circles = [Circle(pos0[i], r, color=AGENT_COLOR, linewidth=0.0, zorder=6) for i in range(n_agent)]
        for c in self.goal_circles + self.agent_circles:
            ax.add_patch(c)
        self.edges = LineCollection([], linewidths=2, alpha=0.5, zorder=3)
        ax.add_collection(self.edges)
        font = dict(size=16, color="k", transform=ax.transAxes)
        self.cost_text = ax.text(0.02, 1.00, "", va="bottom", **font)
        self.unsafe_text = ax.text(0.99, 1.00, "", va="bottom", ha="right", **font)
        self.step_text = ax.text(0.99, 1.04, "", va="bottom", ha="right", **font)
        self.labels = [ax.text(pos0[i, 0], pos0[i, 1], f"{i}", size=20, color="k", ha="center", va="center", clip_on=True,
                               zorder=7) for i in range(n_agent)]
        self._plt = plt

    def artists(self):
        return [*self.agent_circles, *self.goal_circles, self.edges, self.cost_text, self.unsafe_text, self.step_text,
                *self.labels]

    def draw(self, t: int):
        ep, n = self.ep, self.n_agent
        pos = ep.states[t, :, :2]
        for i, c in enumerate(self.agent_circles):
            c.set_center(tuple(pos[i]))
            self.labels[i].set_position(tuple(pos[i]))
        for j, c in enumerate(self.goal_circles):
            c.set_center(tuple(pos[n + j]))
        seg, from_goal = frame_edges(ep, t, n, self.n_goal)
        self.edges.set_segments(list(seg))
        self.edges.set_colors([GOAL_COLOR if g else COMM_COLOR for g in from_goal])
        worst = ep.costs[t].max(axis=0)                        # per component: the worst agent (plot.py cost text)
        lines = [f"{name}: {worst[k]:8.3f}" for k, name in enumerate(self.cost_components[:len(worst)])]
        self.cost_text.set_text("Cost:\n  " + "\n  ".join(lines) + f"\nReward: {ep.rewards[t]:.3f}")
        self.unsafe_text.set_text("Unsafe: {}".format(unsafe_agents(ep, t, self.Ta_is_unsafe).tolist()))
        self.step_text.set_text(f"kk={t:04}")
        return self.artists()

    def close(self):
        self._plt.close(self.fig)


def _write(scene: _Scene, n_frames: int, video_path: pathlib.Path, fps: int = 33) -> pathlib.Path:
    from matplotlib import animation
    video_path = pathlib.Path(video_path)
    video_path.parent.mkdir(parents=True, exist_ok=True)
    anim = animation.FuncAnimation(scene.fig, scene.draw, frames=n_frames, init_func=scene.artists, interval=1000 // fps,
                                   blit=False)
    out, writer = video_path, None
    if video_path.suffix.lower() == ".gif":
        writer = animation.PillowWriter(fps=fps)
    elif not animation.writers.is_available("ffmpeg"):
        out = video_path.with_suffix(".gif")
        writer = animation.PillowWriter(fps=fps)
        print(f"(no ffmpeg binary: writing {out.name} instead of {video_path.name})")
    else:
        writer = animation.FFMpegWriter(fps=fps)
    anim.save(str(out), writer=writer)
    scene.close()
    return out


def _render(rollout, video_path, side_length, dim, n_agent, r, obs_r, cost_components, Ta_is_unsafe, viz_opts, dpi, n_goal,
            index=None, max_frames=None, **kwargs) -> pathlib.Path:
    if dim != 2:
        raise NotImplementedError("only the planar environments of SURVEY §8 are rendered (dim == 2)")
    if viz_opts:
        raise NotImplementedError(f"viz_opts {sorted(viz_opts)}: CBF / Vh overlays are not built")
    ep = episode_from_rollout(rollout, index)
    n_goal = n_agent if n_goal is None else n_goal
    scene = _Scene(ep, float(side_length), n_agent, n_goal, r, obs_r, cost_components, Ta_is_unsafe, dpi)
    T = ep.states.shape[0] if max_frames is None else min(ep.states.shape[0], max_frames)
    return _write(scene, T, video_path)


def render_lidar(rollout, video_path, side_length: float, dim: int, n_agent: int, n_rays: int, r: float,
                 cost_components: Tuple[str, ...], Ta_is_unsafe=None, viz_opts: Optional[dict] = None, dpi: int = 100,
                 n_goal: Optional[int] = None, **kwargs) -> pathlib.Path:
    """dgppo/env/plot.py:468 (LiDAR family: rectangle obstacles, `n_rays` hit nodes per agent)."""
    return _render(rollout, video_path, side_length, dim, n_agent, r, 0.0, cost_components, Ta_is_unsafe, viz_opts, dpi,
                   n_goal, **kwargs)


def render_mpe(rollout, video_path, side_length: float, dim: int, n_agent: int, n_obs: int, r: float, obs_r: float,
               cost_components: Tuple[str, ...], Ta_is_unsafe=None, viz_opts: Optional[dict] = None, dpi: int = 100,
               n_goal: Optional[int] = None, **kwargs) -> pathlib.Path:
    """dgppo/env/plot.py:206 (MPE family: disc obstacles of radius obs_r)."""
    return _render(rollout, video_path, side_length, dim, n_agent, r, obs_r, cost_components, Ta_is_unsafe, viz_opts, dpi,
                   n_goal, **kwargs)
