"""DQN-boosted MPC, batched: the decision loop of the reference's ``src/main.py:93-243`` for B robots per tick.

Pieces (each mirrors one reference item):

* :class:`HintSwitcher`   -- ``src/main_pre.py:27-52``: when does the MPC track the DQN's proposal instead of the
  original reference?  Same state machine, polygon distance / containment written out (shapely is not a dependency).
* :func:`ref_traj_filter` -- ``src/main.py:34-41``.
* :func:`circle_to_rect`  -- ``src/main.py:86-90``.
* :func:`inflate_polygon` -- ``Inflator`` / ``geometry_tools.polygon_inflate`` (``src/main_pre.py:18-24``,
  ``src/pkg_obstacle/geometry_tools.py:21-23``): mitre-join offset, the obstacle list the MPC and the switcher see.
* :class:`BatchedHybrid`  -- the loop: environment status / observation (``rl_env.BatchedRaysEnv``, one HIP kernel),
  Q-network action, 20-step RL reference (``dqn.rl_reference``), constant-velocity obstacle predictions
  (``feeders``), reference switch, one batched MPC solve (``BatchedTracker`` -> ``libmpcgpu.so``).
  ``decision_mode``: 0 pure DQN, 1 pure MPC, 2 hybrid -- as in ``main.py:96-98``.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import rl_env, rl_geometry as rg
from .batched_tracker import BatchedTracker
from .config import MpcConfig
from .dqn import QNetwork, merge_reference, rl_reference
from .feeders import DYN_OBS_SIZE, constant_velocity_prediction

MAX_RUN_STEP = 200  # main.py:30


# ---- geometry the switcher needs -----------------------------------------------------------------------------------
def polygon_distance(polygon: np.ndarray, point) -> float:
    """shapely ``Polygon.distance(Point)``: 0 inside, else the distance to the outline."""
    ring = np.asarray(polygon, dtype=float).reshape(-1, 2)
    if rg.point_in_ring(point, ring):
        return 0.0
    px, py = float(point[0]), float(point[1])
    best = math.inf
    for i in range(len(ring)):
        ax, ay = ring[i]
        bx, by = ring[(i + 1) % len(ring)]
        dx, dy = bx - ax, by - ay
        den = dx * dx + dy * dy
        t = 0.0 if den == 0.0 else min(1.0, max(0.0, ((px - ax) * dx + (py - ay) * dy) / den))
        best = min(best, math.hypot(px - (ax + t * dx), py - (ay + t * dy)))
    return best


def inflate_polygon(polygon: Sequence[Sequence[float]], margin: float, mitre_limit: float = 5.0) -> List[List[float]]:
    """Mitre-join offset of a simple polygon (GEOS ``buffer(margin, join_style=mitre)`` with the default limit 5):
    every vertex moves to the intersection of its two offset edges; a corner sharper than the limit is bevelled at
    ``mitre_limit * margin`` from the vertex.  Local construction, valid while the offset ring stays simple (true for
    the reference's obstacle sets)."""
    ring = rg.orient(polygon, ccw=True)
    n = len(ring)
    out: List[List[float]] = []
    for i in range(n):
        v = ring[i]
        d0 = v - ring[i - 1]
        d1 = ring[(i + 1) % n] - v
        d0, d1 = d0 / np.hypot(*d0), d1 / np.hypot(*d1)
        n0 = np.array([d0[1], -d0[0]])
        n1 = np.array([d1[1], -d1[0]])
        cosang = float(n0 @ n1)
        if 1.0 + cosang < 1e-12:                       # a spike: 180 degree turn
            out += [(v + margin * n0).tolist(), (v + margin * n1).tolist()]
            continue
        mitre = (n0 + n1) / (1.0 + cosang)             # |mitre| = 1 / cos(half turning angle)
        ratio = float(np.hypot(*mitre))
        turn = d0[0] * d1[1] - d0[1] * d1[0]
        if ratio <= mitre_limit or turn * margin < 0:  # inside turns always collapse to the intersection
            out.append((v + margin * mitre).tolist())
        else:                                          # limited mitre: bevel at mitre_limit * |margin| on the bisector
            bis = mitre / ratio
            mid = v + margin * mitre_limit * bis
            perp = np.array([-bis[1], bis[0]])
            # the bevel's end points lie on the two offset edges
            h0 = ((v + margin * n0 - mid) @ n0) / (perp @ n0) if abs(perp @ n0) > 1e-15 else 0.0
            out += [(mid + h0 * perp).tolist(), (mid - h0 * perp).tolist()]
    return out


def circle_to_rect(pos, radius: float = DYN_OBS_SIZE) -> List[List[float]]:
    return [[pos[0] - radius, pos[1] - radius], [pos[0] + radius, pos[1] - radius],
            [pos[0] + radius, pos[1] + radius], [pos[0] - radius, pos[1] + radius]]


def ref_traj_filter(original: np.ndarray, new: np.ndarray, decay: float = 1.0) -> np.ndarray:
    """Blend of the two references with a weight that is squared after every row (1 stays 1: the proposal is taken
    as it is) and cut to 0 below 1e-2."""
    filtered = np.array(original, dtype=float, copy=True)
    for i in range(filtered.shape[0]):
        filtered[i, :] = (1 - decay) * filtered[i, :] + decay * new[i, :]
        decay *= decay
        if decay < 1e-2:
            decay = 0.0
    return filtered


class HintSwitcher:
    """Switch to the DQN's proposal when the original reference runs through an obstacle that is closer than
    ``max_switch_distance``; switch back after the robot has been farther than ``min_detach_distance`` from an
    obstacle for more than ``min_detach_steps`` calls.  The loop order and the early return are the reference's."""

    def __init__(self, max_switch_distance: float, min_detach_distance: float, min_detach_steps: float = 5):
        self.switch_distance = max_switch_distance
        self.detach_distance = min_detach_distance
        self.detach_steps = min_detach_steps
        self.detach_cnt = 0
        self.switch_on = False

    def switch(self, current_position, original_traj, new_traj, obstacle_list) -> bool:
        counted = False
        rings = [np.asarray(ob, dtype=float).reshape(-1, 2) for ob in obstacle_list]
        dists = [polygon_distance(r, current_position) for r in rings]   # does not depend on the trajectory row
        for old_pos, _new_pos in zip(original_traj, new_traj):
            for ring, dist in zip(rings, dists):
                if rg.point_in_ring(old_pos[:2], ring):
                    if dist < self.switch_distance and not self.switch_on:
                        self.switch_on = True
                        return self.switch_on
                elif dist > self.detach_distance and self.switch_on:
                    if self.detach_cnt > self.detach_steps:
                        self.switch_on = False
                        self.detach_cnt = 0
                    elif not counted:
                        self.detach_cnt += 1
                        counted = True
        return self.switch_on


def filter_weights(n_rows: int, decay: float = 1.0) -> np.ndarray:
    """Row weights of :func:`ref_traj_filter` (they do not depend on the data)."""
    w = np.empty(n_rows)
    for i in range(n_rows):
        w[i] = decay
        decay *= decay
        if decay < 1e-2:
            decay = 0.0
    return w


def pad_polygons(polygons: Sequence[Sequence[Sequence[float]]], n_vertices: int) -> np.ndarray:
    """[O, n_vertices, 2]: every ring padded by repeating its last vertex (zero-length edges change neither the
    even-odd test nor the distance)."""
    out = np.zeros((len(polygons), n_vertices, 2))
    for i, poly in enumerate(polygons):
        ring = np.asarray(poly, dtype=float).reshape(-1, 2)
        out[i, :len(ring)] = ring
        out[i, len(ring):] = ring[-1]
    return out


def points_in_polygons(points: np.ndarray, polygons: np.ndarray) -> np.ndarray:
    """Even-odd rule, batched: points [B, R, 2], polygons [B, O, V, 2] -> [B, R, O] bool."""
    a = polygons[:, None]                               # [B, 1, O, V, 2]
    b = np.roll(polygons, -1, axis=2)[:, None]
    px, py = points[:, :, None, None, 0], points[:, :, None, None, 1]
    straddle = (a[..., 1] > py) != (b[..., 1] > py)
    dy = np.where(straddle, b[..., 1] - a[..., 1], 1.0)
    cross = px < (b[..., 0] - a[..., 0]) * (py - a[..., 1]) / dy + a[..., 0]
    return (np.sum(straddle & cross, axis=-1) & 1).astype(bool)


def polygon_distances(points: np.ndarray, polygons: np.ndarray) -> np.ndarray:
    """shapely ``Polygon.distance(Point)``, batched: points [B, 2], polygons [B, O, V, 2] -> [B, O]."""
    a = polygons
    d = np.roll(polygons, -1, axis=2) - a
    rel = points[:, None, None, :] - a
    den = np.sum(d * d, axis=-1)
    t = np.clip(np.where(den > 0.0, np.sum(rel * d, axis=-1) / np.where(den > 0.0, den, 1.0), 0.0), 0.0, 1.0)
    gap = rel - t[..., None] * d
    dist = np.sqrt(np.sum(gap * gap, axis=-1)).min(axis=-1)
    inside = points_in_polygons(points[:, None, :], polygons)[:, 0]
    return np.where(inside, 0.0, dist)


class BatchedHintSwitcher:
    """:class:`HintSwitcher` for B robots at once: the same sequential scan over (trajectory row, obstacle) pairs, each
    step applied to all robots as array operations (early return = a per-robot mask)."""

    def __init__(self, n_robots: int, max_switch_distance: float, min_detach_distance: float, min_detach_steps: float = 5):
        self.B = n_robots
        self.switch_distance, self.detach_distance, self.detach_steps = max_switch_distance, min_detach_distance, min_detach_steps
        self.reset()

    def reset(self):
        self.detach_cnt = np.zeros(self.B, dtype=np.int64)
        self.switch_on = np.zeros(self.B, dtype=bool)

    def switch(self, positions: np.ndarray, original: np.ndarray, polygons: np.ndarray, valid: np.ndarray,
               live: Optional[np.ndarray] = None) -> np.ndarray:
        """positions [B, 2]; original [B, R, >= 2] (rows of the original reference); polygons [B, O, V, 2] with
        ``valid`` [B, O] marking real obstacles; ``live`` [B]: robots whose switcher is consulted this tick."""
        contains = points_in_polygons(np.asarray(original)[..., :2], polygons)      # [B, R, O]
        dist = polygon_distances(np.asarray(positions, dtype=float), polygons)      # [B, O]
        on, cnt = self.switch_on, self.detach_cnt
        returned = np.zeros(self.B, dtype=bool) if live is None else ~np.asarray(live, dtype=bool)
        counted = np.zeros(self.B, dtype=bool)
        near, far = dist < self.switch_distance, dist > self.detach_distance
        for r in range(contains.shape[1]):
            for o in range(contains.shape[2]):
                act = valid[:, o] & ~returned
                c = contains[:, r, o]
                trig = act & c & near[:, o] & ~on
                on = on | trig
                returned = returned | trig
                e = act & ~c & far[:, o] & on & ~trig
                reset = e & (cnt > self.detach_steps)
                inc = e & ~reset & ~counted
                on = on & ~reset
                cnt = np.where(reset, 0, cnt + inc)
                counted = counted | inc
        self.switch_on, self.detach_cnt = on, cnt
        return on.copy()


# ---- the batched decision loop ----------------------------------------------------------------------------------------
class BatchedHybrid:
    """B robots, each in its own copy of a scene.  ``scenes[i]``: dict with ``boundary``, ``static`` (polygons),
    ``dynamic`` (``rl_env.periodic_obstacle`` arguments), ``start`` (x, y, theta, v, w), ``goal`` (x, y), ``path``.

    One tick = one environment kernel launch + one Q-network forward + one batched MPC solve."""

    def __init__(self, config: MpcConfig, scenes: Sequence[Dict], q_net: QNetwork, decision_mode: int = 2,
                 device: int = 0, inflate_margin: float = 0.8, switcher=(10, 2, 10), tracker: Optional[BatchedTracker] = None):
        import torch
        if decision_mode not in (0, 1, 2):
            raise ValueError("decision_mode: 0 pure DQN, 1 pure MPC, 2 hybrid")
        self._torch = torch
        self.config, self.mode, self.B = config, decision_mode, len(scenes)
        self.scenes = list(scenes)
        self.maps = [rl_env.make_map(s["boundary"], s["static"], s["dynamic"], s["start"], s["goal"], s["path"])
                     for s in scenes]
        self.env = rl_env.BatchedRaysEnv(self.maps, device=device, time_step=0.2)   # gym.make default (environment.py:53)
        self.q_net = q_net.to(self.env.device)
        self.tracker = tracker if tracker is not None else BatchedTracker(config, self.B, device=device)
        self.inflated = [[inflate_polygon(poly, inflate_margin) for poly in s["static"]] for s in scenes]
        self.switcher = BatchedHintSwitcher(self.B, *switcher)
        n_dyn = max(len(s["dynamic"]) for s in scenes)
        self._n_static = np.array([len(s["static"]) for s in scenes])
        self._n_dynamic = np.array([len(s["dynamic"]) for s in scenes])
        vmax = max([4] + [len(p) for polys in self.inflated for p in polys])
        omax = int(self._n_static.max()) + n_dyn
        self._polygons = np.zeros((self.B, omax, vmax, 2))        # static (inflated) first, this tick's discs after them
        self._poly_valid = np.zeros((self.B, omax), dtype=bool)
        for b, polys in enumerate(self.inflated):
            if polys:
                self._polygons[b, :len(polys)] = pad_polygons(polys, vmax)
            self._poly_valid[b, :len(polys)] = True
            self._poly_valid[b, self._n_static.max():self._n_static.max() + self._n_dynamic[b]] = True
        self.reset()

    def reset(self):
        cfg = self.config
        self.obs = self.env.reset()
        for i, s in enumerate(self.scenes):
            start = np.asarray(s["start"], dtype=float)
            self.tracker.initialization(i, start[:3], np.array([s["goal"][0], s["goal"][1], 0.0]), s["path"])
            self.tracker.update_static_constraints(i, self.inflated[i])
        self.switcher.reset()
        self.last_dyn = None
        self.done = np.zeros(self.B, dtype=bool)
        self.success = np.zeros(self.B, dtype=bool)
        self.collided = np.zeros(self.B, dtype=bool)
        self.steps = np.zeros(self.B, dtype=int)
        self.switch_on = np.zeros(self.B, dtype=bool)
        self.switch_ticks = np.zeros(self.B, dtype=int)   # ticks on which the MPC tracked the DQN's proposal
        self.t = 0

    def dynamic_positions(self) -> np.ndarray:
        """[B, Kmax, 2]: current key-frame positions of every environment's dynamic obstacles (main.py:127); rows
        beyond an environment's own count stay 0."""
        clock = self.env.state[:, 5].cpu().numpy()
        out = np.zeros((self.B, int(self._n_dynamic.max()) if self.B else 0, 2))
        for b, m in enumerate(self.maps):
            for j, ob in enumerate(m["obstacles"][self._n_static[b]:]):
                out[b, j] = rl_env.keyframe_pose(ob, clock[b])[:2]
        return out

    def _flat(self, obs):
        return self._torch.cat([obs["external"], obs["internal"]], dim=1)

    def _mark(self, name: str) -> None:
        """Per-tick breakdown (``self.phase_seconds``; enabled by ``self.profile = True``): wall time between marks with the
        device drained at every mark, so that a phase is charged with the kernels it launched."""
        if not getattr(self, "profile", False):
            return
        import time
        self._torch.cuda.synchronize()
        now = time.perf_counter()
        if not hasattr(self, "phase_seconds"):
            self.phase_seconds = {}
        if name != "start":
            self.phase_seconds[name] = self.phase_seconds.get(name, 0.0) + now - self._last_mark
        self._last_mark = now

    def tick(self) -> Dict[str, np.ndarray]:
        torch, cfg, env, trk = self._torch, self.config, self.env, self.tracker
        self._mark("start")
        dyn_now = self.dynamic_positions()
        if self.last_dyn is None:
            self.last_dyn = dyn_now
        kmax = dyn_now.shape[1]
        if kmax:                                             # est_dyn_obs_positions (main.py:77-85) for every robot
            has = (np.arange(kmax)[None, :] < self._n_dynamic[:, None])[:, :, None, None]
            preds = np.where(has, constant_velocity_prediction(self.last_dyn, dyn_now, steps=cfg.N_hor), 0.0)
        self.last_dyn = dyn_now
        live = ~self.done
        self._mark("obstacle predictions (host)")

        if self.mode == 0:                                   # pure DQN: main.py:139-152
            actions = self.q_net.greedy_actions(self._flat(self.obs))
            self.obs, _, term, trunc, info = env.step(actions)
            trk.states[:] = env.agent_state[:, :3].cpu().numpy()
            chosen = None
        else:
            # the environment follows the tracker: position / heading from the MPC state, speeds from its last action
            st = np.concatenate([trk.states, trk.last_actions], axis=1)
            env.set_agent_state(st)
            if self.mode == 1:                               # main.py:154-157: env.step(0) "just for ... status"
                self.obs, _, term, trunc, info = env.step(torch.zeros(self.B, dtype=torch.int32))
                self._mark("environment kernel")
                chosen = trk.local_refs()
                self._mark("local reference (host)")
            else:                                            # main.py:176-214
                actions = self.q_net.greedy_actions(self._flat(self.obs)).cpu().numpy()
                self._mark("Q-network")
                env.state[:, 5] += env.time_step             # step_obstacles()
                self.obs = env.observe()                     # update_status() + get_observation()
                term = env.terminated.bool()
                info = {"success": env.flags[:, 2]}
                self._mark("environment kernel")
                rl_ref, _ = rl_reference(env.agent_state.cpu().numpy(), actions, cfg.ts, steps=20, ref_speed=1.0)
                original = trk.local_refs()
                proposal = merge_reference(rl_ref[:, :cfg.N_hor], original)
                w = filter_weights(cfg.N_hor, 1.0)[None, :, None]      # ref_traj_filter(decay=1): the proposal as it is
                filtered = (1.0 - w) * original + w * proposal
                if kmax:                                             # circle_to_rect of every disc (main.py:129)
                    r = DYN_OBS_SIZE
                    corners = np.array([[-r, -r], [r, -r], [r, r], [-r, r]])
                    s0 = int(self._n_static.max())
                    rects = dyn_now[:, :, None, :] + corners[None, None]
                    self._polygons[:, s0:s0 + kmax, :4] = rects
                    self._polygons[:, s0:s0 + kmax, 4:] = rects[:, :, 3:4]
                on = self.switcher.switch(trk.states[:, :2], original, self._polygons, self._poly_valid, live)
                self.switch_on = on & live
                self.switch_ticks += self.switch_on
                chosen = np.where(self.switch_on[:, None, None], filtered, original)
                self._mark("RL reference + switch (host)")
            if kmax:
                trk.set_dynamic_constraints(preds)
            trk.active &= live
            trk.step(refs=chosen)
            self._mark("parameter assembly + batched MPC solve")
            # get_action returns None once the tracker's own termination test fires (interface_mpc.py:83-85)
            self.done |= live & ~trk.active
        term = term.cpu().numpy().astype(bool)
        flags = env.flags.cpu().numpy()
        self.success |= live & flags[:, 2]
        self.collided |= live & (flags[:, 0] | flags[:, 1])
        self.done |= live & term
        self.steps += live
        self.t += 1
        self._mark("bookkeeping (host)")
        return dict(done=self.done.copy(), success=self.success.copy(), collided=self.collided.copy(),
                    switch_on=self.switch_on.copy(), states=trk.states.copy())

    def run(self, max_steps: int = MAX_RUN_STEP, record: bool = False) -> Dict[str, np.ndarray]:
        """Run until every robot is done (or ``max_steps``).  ``record=True`` also returns what the reference's
        ``Metrics.add_trial_result`` consumes per robot (``main_evaluation.py:255-266``): tick times [ms], (v, w) history,
        traversed positions and the global reference trajectory (see ``metrics.Metrics.add_batch``)."""
        import time
        out = None
        tick_ms = [[] for _ in range(self.B)]
        actions = [[(float(s["start"][3]), float(s["start"][4]))] for s in self.scenes]
        positions = [[(float(s["start"][0]), float(s["start"][1]))] * 2 for s in self.scenes]   # update_status appends twice at reset
        for _ in range(max_steps):
            live = ~self.done
            t0 = time.perf_counter()
            out = self.tick()
            dt = 1e3 * (time.perf_counter() - t0)
            if record:
                st = self.env.agent_state.cpu().numpy()
                for b in np.nonzero(live)[0]:
                    tick_ms[b].append(dt)
                    actions[b].append((float(st[b, 3]), float(st[b, 4])))
                    positions[b].append((float(st[b, 0]), float(st[b, 1])))
            if out["done"].all():
                break
        res = dict(out, steps=self.steps.copy(), switch_ticks=self.switch_ticks.copy(),
                   progress=self.env.path_progress.cpu().numpy())
        if record:
            res["record"] = dict(tick_ms=tick_ms, success=self.success.copy(), actions=actions, positions=positions,
                                 ref_traj=[np.array(t) for t in self.tracker.ref_trajs])
        return res
