"""TEST INFRASTRUCTURE -- CPU restatement of the reference's DRL environment step.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(``trajtrack_mpcndqn_rlboost_amd.rl_env`` -> ``mpcgpu_env_step_dev`` in libmpcgpu.so) never does.

What is restated, one scalar environment at a time, plain Python / numpy float64 (paths relative to
/root/reference/src/pkg_dqn/environment/):

* robot kinematics with the 3 x 3 discrete acceleration actions          agent.py:97-139
* cyclic key-frame animation of the obstacles                            obstacle.py:52-105
* status flags (collision with padded obstacles / padded boundary, goal)  environment.py:96-122
* sector + ray observation with one-step memory                          components/ext_obsv_sector_and_ray.py:31-81
* speed / angular velocity / path-sample / path-corner observations       components/int_obsv_*.py
* reward R1 (collision, cross-track, reach-goal, excessive speed, progress) variants/rays_reward1.py:26-39,
                                                                            components/reward_*.py
* shapely's LineString.project / interpolate (first closest segment wins, clamped to the line)

Pinning: tests/golden/env_rays_traces.npz holds traces produced by the reference's OWN environment / component code
(imported from /root/reference by tests/golden/make_env_fixtures.py) running on a small geometry shim in place of
shapely, which is not installed here.  The shim's geometric primitives are the ones in this file, so the traces pin the
environment logic (ordering, flags, memory, normalisation, rewards), not GEOS itself: the padded outlines come from
``rl_geometry.buffer_polygon`` (GEOS' fillet rule restated) -- **parity with GEOS' buffer is unpinned**.

The geometry here is written differently from the HIP kernel on purpose -- the reference's own 1000 m sector triangle
with a generic Cyrus-Beck clip of every outline edge, a 2 x 2 linear solve for the ray hits and a winding-number
inside test, versus the kernel's two wedge half-planes, closed-form cross products and crossing-number parity -- so
agreement between the two is a meaningful check.  (A Sutherland-Hodgman clip of the whole outline was tried first and
dropped: for non-convex outlines it leaves zero-width slivers that can reach the sector apex and fake a distance 0.)
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import numpy as np

L_SECTOR = 1000.0  # ext_obsv_sector_and_ray.py:32


# ------------------------------------------------------------------------------------------------------------------
# geometric primitives
# ------------------------------------------------------------------------------------------------------------------
def winding_number(pt, ring: np.ndarray) -> int:
    x, y = float(pt[0]), float(pt[1])
    wn = 0
    n = len(ring)
    for i in range(n):
        x0, y0 = ring[i]
        x1, y1 = ring[(i + 1) % n]
        is_left = (x1 - x0) * (y - y0) - (x - x0) * (y1 - y0)
        if y0 <= y:
            if y1 > y and is_left > 0:
                wn += 1
        elif y1 <= y and is_left < 0:
            wn -= 1
    return wn


def point_segment_distance(pt, a, b) -> float:
    ax, ay = a
    bx, by = b
    dx, dy = bx - ax, by - ay
    den = dx * dx + dy * dy
    t = 0.0 if den == 0.0 else ((pt[0] - ax) * dx + (pt[1] - ay) * dy) / den
    t = min(1.0, max(0.0, t))
    return math.hypot(pt[0] - (ax + t * dx), pt[1] - (ay + t * dy))


def clip_segment_convex(a, b, clip_ccw: np.ndarray):
    """Cyrus-Beck: the part of segment ab inside the convex CCW ring, or None."""
    t0, t1 = 0.0, 1.0
    dx, dy = b[0] - a[0], b[1] - a[1]
    m = len(clip_ccw)
    for i in range(m):
        c0, c1 = clip_ccw[i], clip_ccw[(i + 1) % m]
        ex, ey = c1[0] - c0[0], c1[1] - c0[1]
        f0 = ex * (a[1] - c0[1]) - ey * (a[0] - c0[0])   # >= 0 inside
        df = ex * dy - ey * dx
        if df == 0.0:
            if f0 < 0:
                return None
            continue
        t = -f0 / df
        if df > 0:
            t0 = max(t0, t)
        else:
            t1 = min(t1, t)
        if t0 > t1:
            return None
    return (a[0] + t0 * dx, a[1] + t0 * dy), (a[0] + t1 * dx, a[1] + t1 * dy)


def ray_segment_hit(o, d, a, b) -> float:
    """Distance s >= 0 along the unit direction d from o to segment ab (inf when they do not meet)."""
    mat = np.array([[d[0], a[0] - b[0]], [d[1], a[1] - b[1]]], dtype=np.float64)
    if abs(np.linalg.det(mat)) < 1e-300:
        return math.inf
    s, t = np.linalg.solve(mat, np.array([a[0] - o[0], a[1] - o[1]], dtype=np.float64))
    if s >= 0.0 and 0.0 <= t <= 1.0:
        return float(s)
    return math.inf


# ------------------------------------------------------------------------------------------------------------------
# shapely.LineString.project / interpolate
# ------------------------------------------------------------------------------------------------------------------
def path_lengths(path: np.ndarray) -> np.ndarray:
    """Cumulative length at every node, summed sequentially in float64 (the order GEOS and the corner observation's
    ``while`` loop both use)."""
    cum = np.zeros(len(path))
    for i in range(1, len(path)):
        cum[i] = cum[i - 1] + math.sqrt((path[i][0] - path[i - 1][0]) ** 2 + (path[i][1] - path[i - 1][1]) ** 2)
    return cum


def path_project(path: np.ndarray, cum: np.ndarray, pt) -> float:
    best, best_s = math.inf, 0.0
    for i in range(len(path) - 1):
        ax, ay = path[i]
        dx, dy = path[i + 1][0] - ax, path[i + 1][1] - ay
        den = dx * dx + dy * dy
        t = 0.0 if den == 0.0 else ((pt[0] - ax) * dx + (pt[1] - ay) * dy) / den
        t = min(1.0, max(0.0, t))
        dist = math.hypot(pt[0] - (ax + t * dx), pt[1] - (ay + t * dy))
        if dist < best:  # strict: the first closest segment wins
            best, best_s = dist, cum[i] + t * math.sqrt(den)  # cum[i + 1] == cum[i] + sqrt(den) exactly
    return best_s


def path_interpolate(path: np.ndarray, cum: np.ndarray, s: float):
    if s <= 0.0:
        return float(path[0][0]), float(path[0][1])
    if s >= cum[-1]:
        return float(path[-1][0]), float(path[-1][1])
    for i in range(len(path) - 1):
        if s < cum[i + 1]:
            seg = cum[i + 1] - cum[i]
            t = (s - cum[i]) / seg
            return (float(path[i][0] + t * (path[i + 1][0] - path[i][0])),
                    float(path[i][1] + t * (path[i + 1][1] - path[i][1])))
    return float(path[-1][0]), float(path[-1][1])


def normalize_distance(d: float, max_distance: float = 10.0) -> float:  # components/utils.py:10-15
    if math.isinf(d):
        return 1.0
    return 2.0 / (1.0 + math.exp(-2.0 * d / max_distance)) - 1.0


# ------------------------------------------------------------------------------------------------------------------
# obstacle animation  (obstacle.py:52-105)
# ------------------------------------------------------------------------------------------------------------------
def keyframe_pose(time_steps: Sequence[float], keyframes: Sequence[Sequence[float]], interp: str, offset: float,
                  time: float):
    """(x, y, rotation) at ``time``; ``keyframes[i] = (x, y, rotation)``, ``len(time_steps) == len(keyframes) + 1``."""
    length = float(sum(time_steps))
    tm = (time + offset) % length
    t = 0.0
    nk = len(keyframes)
    for i in range(nk):
        t += time_steps[i]
        if t <= tm < t + time_steps[i + 1]:
            x = (tm - t) / time_steps[i + 1]
            alpha = (1.0 - math.cos(x * math.pi)) / 2.0 if interp == "cosine" else x
            k0, k1 = keyframes[i], keyframes[(i + 1) % nk]
            return tuple(k0[j] * (1.0 - alpha) + k1[j] * alpha for j in range(3))
    return tuple(keyframes[-1])


def transform(nodes: np.ndarray, pose) -> np.ndarray:
    c, s = math.cos(pose[2]), math.sin(pose[2])
    rot = np.array([[c, -s], [s, c]])
    return np.asarray(pose[:2]) + (rot @ nodes.T).T   # obstacle.py:174-188


# ------------------------------------------------------------------------------------------------------------------
# the environment
# ------------------------------------------------------------------------------------------------------------------
class RobotSpec:  # agent.py:7-16
    RADIUS = 0.5
    SPEED_MIN, SPEED_MAX = -0.5, 1.5
    ANGULAR_VELOCITY_MIN, ANGULAR_VELOCITY_MAX = -0.5, 0.5
    ACCELERATION_MIN, ACCELERATION_MAX = -1.0, 1.0
    ANGULAR_ACCELERATION_MIN, ANGULAR_ACCELERATION_MAX = -3.0, 3.0


class OracleRaysEnv:
    """``TrajectoryPlannerEnvironmentRaysReward1`` for ONE environment with a given map and reference path.

    ``spec`` keys: ``start`` (x, y, theta, v, w), ``goal`` (x, y), ``path`` [P][2], ``boundary_padded`` ring,
    ``obstacles``: list of dicts ``padded_nodes`` (body frame ring), ``time_steps``, ``keyframes`` [(x, y, rot)],
    ``interp`` ("linear" | "cosine"), ``offset``."""

    def __init__(self, spec: Dict, time_step: float = 0.2, num_segments: int = 8, corner_samples: int = 3,
                 sample_offset: float = 0.0, collision_factor: float = 4.0, reach_goal_factor: float = 3.0,
                 cross_track_factor: float = 0.05, reference_speed: float = 1.5 * 0.8, path_progress_factor: float = 2.0):
        self.spec = spec
        self.ts = time_step
        self.nseg = num_segments
        self.ncorner = corner_samples
        self.sample_offset = sample_offset
        self.f_coll, self.f_goal, self.f_cte = collision_factor, reach_goal_factor, cross_track_factor
        self.f_speed, self.ref_speed, self.f_prog = 2.0 * path_progress_factor, reference_speed, path_progress_factor
        self.path = np.asarray(spec["path"], dtype=np.float64)
        self.cum = path_lengths(self.path)
        self.boundary = np.asarray(spec["boundary_padded"], dtype=np.float64)
        self.old_obs = np.zeros(4 * num_segments, dtype=np.float32)  # never cleared by reset (the component has no reset())
        self.reset()

    # ---- environment.py:166-186 (map and path are inputs here) -----------------------------------------------------
    def reset(self):
        self.state = np.asarray(self.spec["start"], dtype=np.float64).copy()
        self.time = 0.0
        self.collided_obstacle = self.collided_boundary = self.collided = self.reached_goal = False
        self._update_status()
        self.last_progress = 0.0  # reward_path_progress.py:11-12
        return self.observation()

    def obstacle_rings(self) -> List[np.ndarray]:
        rings = []
        for ob in self.spec["obstacles"]:
            pose = keyframe_pose(ob["time_steps"], ob["keyframes"], ob["interp"], ob["offset"], self.time)
            rings.append(transform(np.asarray(ob["padded_nodes"], dtype=np.float64), pose))
        return rings

    def _update_status(self):  # environment.py:96-122
        pos = self.state[:2]
        rings = self.obstacle_rings()
        self.collided_obstacle |= any(winding_number(pos, r) != 0 for r in rings)
        self.collided_boundary |= winding_number(pos, self.boundary) == 0
        self.collided |= self.collided_obstacle or self.collided_boundary
        goal = self.spec["goal"]
        self.reached_goal |= math.hypot(goal[0] - pos[0], goal[1] - pos[1]) < RobotSpec.RADIUS
        self.progress = path_project(self.path, self.cum, pos)

    # ---- observations ------------------------------------------------------------------------------------------------
    def _rel(self, point):
        dx, dy = point[0] - self.state[0], point[1] - self.state[1]
        rel = math.atan2(dy, dx) - self.state[2]
        return [math.cos(rel), math.sin(rel), normalize_distance(math.hypot(dx, dy))]

    def internal_obs(self) -> np.ndarray:
        v, w = self.state[3], self.state[4]
        obs = [2.0 * (v - RobotSpec.SPEED_MIN) / (RobotSpec.SPEED_MAX - RobotSpec.SPEED_MIN) - 1.0,
               # quirk kept: the angular VELOCITY is normalised with the angular ACCELERATION limits
               # (int_obsv_angular_velocity.py:13-19)
               2.0 * (w - RobotSpec.ANGULAR_ACCELERATION_MIN) /
               (RobotSpec.ANGULAR_ACCELERATION_MAX - RobotSpec.ANGULAR_ACCELERATION_MIN) - 1.0]
        obs += self._rel(path_interpolate(self.path, self.cum, self.progress + self.sample_offset))
        # int_obsv_reference_path_corner.py:25-45
        length, i = 0.0, 0
        while length < self.progress and i < len(self.path) - 1:
            length += math.sqrt((self.path[i + 1][0] - self.path[i][0]) ** 2 + (self.path[i + 1][1] - self.path[i][1]) ** 2)
            i += 1
        for _ in range(self.ncorner):
            i = min(len(self.path) - 1, i)
            obs += self._rel(self.path[i])
            i += 1
        return np.asarray(obs, dtype=np.float32)

    def sector_ray_distances(self):
        pos, theta = self.state[:2], self.state[2]
        rings = self.obstacle_rings()
        width = 2.0 * math.pi / self.nseg
        sectors = np.full(self.nseg, math.inf)
        rays = np.full(self.nseg, math.inf)
        inside = [winding_number(pos, r) != 0 for r in rings]
        nb = len(self.boundary)
        for i in range(self.nseg):
            ang = theta + i * width
            a1, a2 = ang - width / 2.0, ang + width / 2.0
            tri = np.array([[pos[0], pos[1]],
                            [pos[0] + L_SECTOR * math.cos(a1), pos[1] + L_SECTOR * math.sin(a1)],
                            [pos[0] + L_SECTOR * math.cos(a2), pos[1] + L_SECTOR * math.sin(a2)]])
            d = (math.cos(ang), math.sin(ang))
            for ring, ins in zip(rings, inside):
                if ins:  # the apex itself belongs to the intersection
                    sectors[i] = 0.0
                    rays[i] = 0.0
                    continue
                # robot outside the outline: the closest point of (outline ∩ sector) lies on the outline's edges
                for k in range(len(ring)):
                    piece = clip_segment_convex(ring[k], ring[(k + 1) % len(ring)], tri)
                    if piece is not None:
                        sectors[i] = min(sectors[i], point_segment_distance(pos, piece[0], piece[1]))
                    hit = ray_segment_hit(pos, d, ring[k], ring[(k + 1) % len(ring)])
                    if hit <= L_SECTOR:
                        rays[i] = min(rays[i], hit)
            for k in range(nb):  # boundary: a line string, no interior
                piece = clip_segment_convex(self.boundary[k], self.boundary[(k + 1) % nb], tri)
                if piece is not None:
                    sectors[i] = min(sectors[i], point_segment_distance(pos, piece[0], piece[1]))
                hit = ray_segment_hit(pos, d, self.boundary[k], self.boundary[(k + 1) % nb])
                if hit <= L_SECTOR:
                    rays[i] = min(rays[i], hit)
        return sectors, rays

    def external_obs(self) -> np.ndarray:
        n = self.nseg
        sectors, rays = self.sector_ray_distances()
        obs = np.zeros(4 * n, dtype=np.float32)
        obs[:n] = [normalize_distance(x) for x in sectors]
        obs[n:2 * n] = [normalize_distance(x) for x in rays]
        obs[2 * n:3 * n] = self.old_obs[:n]
        obs[3 * n:] = self.old_obs[n:2 * n]
        self.old_obs = obs
        return obs

    def observation(self) -> Dict[str, np.ndarray]:
        return {"internal": self.internal_obs(), "external": self.external_obs()}

    # ---- step (environment.py:199-213) ---------------------------------------------------------------------------------
    def step_agent(self, action: int):  # agent.py:97-139
        s, ts = self.state, self.ts
        if action // 3 == 0:
            s[3] += ts * RobotSpec.ACCELERATION_MAX
        if action // 3 == 2:
            s[3] += ts * RobotSpec.ACCELERATION_MIN
        if action % 3 == 0:
            s[4] += ts * RobotSpec.ANGULAR_ACCELERATION_MAX
        if action % 3 == 2:
            s[4] += ts * RobotSpec.ANGULAR_ACCELERATION_MIN
        s[3] = min(RobotSpec.SPEED_MAX, max(RobotSpec.SPEED_MIN, s[3]))
        s[4] = min(RobotSpec.ANGULAR_VELOCITY_MAX, max(RobotSpec.ANGULAR_VELOCITY_MIN, s[4]))
        s[2] += ts * s[4]
        s[0] += ts * s[3] * math.cos(s[2])
        s[1] += ts * s[3] * math.sin(s[2])

    def reward(self) -> float:
        r = -self.f_coll if self.collided else 0.0
        cp = path_interpolate(self.path, self.cum, self.progress)
        cte = math.hypot(self.state[0] - cp[0], self.state[1] - cp[1])
        r += -self.ts * self.f_cte * cte ** 2
        r += self.f_goal if self.reached_goal else 0.0
        err = math.copysign(1.0, self.ref_speed) * (self.state[3] - self.ref_speed)
        r += -self.ts * self.f_speed * max(0.0, err)
        r += self.f_prog * (self.progress - self.last_progress)
        self.last_progress = self.progress
        return r

    def step(self, action: Optional[int]):
        """``action=None``: observe only (``set_agent_state`` + ``update_status`` + ``get_observation``, main.py:181-189)."""
        if action is not None:
            self.time += self.ts
            self.step_agent(int(action))
        self._update_status()
        obs = self.observation()
        rew = self.reward() if action is not None else 0.0
        return obs, rew, bool(self.collided or self.reached_goal), {"success": self.reached_goal}
