"""Batched DRL environment: host side of ``mpcgpu_env_step_dev``.

Mirrors ``TrajectoryPlannerEnvironmentRaysReward1`` of the reference (``src/pkg_dqn/environment/variants/
rays_reward1.py:7-43`` on top of ``environment.py:26-213``) for B environments at once: ``reset`` / ``step`` /
``set_agent_state`` + ``observe`` with the reference's observation dictionary (``internal`` [B, 14], ``external``
[B, 32], float32), reward and termination flags.  All per-step work -- robot and obstacle motion, collision and goal
flags, sector / ray observation with memory, path observations, reward -- is ONE HIP kernel launch
(``csrc/envgpu.hip``); this module only prepares the maps once (padded outlines, key frames, path lengths -> one record
of doubles per environment) and keeps the state tensors.  torch is used for device memory only.

Maps and reference paths are inputs (the reference gets the path from ``extremitypathfinder``, a third-party A*;
``environment.py:124-147``).  There is no CPU path: without the built library and a HIP device construction raises.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import rl_geometry as rg
from .solver import MpcGpuError, load_library

STATE_DOUBLES = 32
N_INTERNAL = 14
N_EXTERNAL = 32
HDR = 16

# MobileRobotSpecification (agent.py:7-16)
ROBOT = dict(radius=0.5, speed_min=-0.5, speed_max=1.5, angvel_min=-0.5, angvel_max=0.5, acc_min=-1.0, acc_max=1.0,
             angacc_min=-3.0, angacc_max=3.0)


def _f32(a) -> np.ndarray:
    """Round through float32 the way the reference's ``np.float32`` node / key-frame arrays do."""
    return np.asarray(a, dtype=np.float32).astype(np.float64)


def static_obstacle(nodes: Sequence[Sequence[float]], radius: float = ROBOT["radius"]) -> Dict:
    """``Obstacle.create_mpc_static`` (obstacle.py:189-191): polygon padded by the robot radius, no motion."""
    padded = _f32(rg.buffer_polygon(_f32(rg.orient(nodes)), radius))
    return dict(padded_nodes=padded, time_steps=[0.0, 1.0], keyframes=[(0.0, 0.0, 0.0)], interp="linear", offset=0.0)


def periodic_obstacle(p1, p2, freq: float, rx: float, ry: float, angle: float, corners: int = 12,
                      radius: float = ROBOT["radius"]) -> Dict:
    """``Obstacle.create_mpc_dynamic`` (obstacle.py:193-201): an ellipse-like polygon oscillating between ``p1`` and
    ``p2`` with cosine easing (``Animation.periodic``, obstacle.py:97-105)."""
    padded = _f32(rg.buffer_polygon(_f32(rg.ellipse_nodes(rx, ry, corners)), radius))
    step = math.pi / freq if freq != 0 else 1.0
    p1, p2 = _f32(p1), _f32(p2)
    # Reference quirk, kept: the loop that builds the nodes re-uses the name ``angle`` (obstacle.py:196-198), so the
    # key frames are created with the LAST node angle 2 pi (corners - 1) / corners, not with the caller's ``angle``.
    rot = 2.0 * math.pi * (corners - 1) / corners if corners > 0 else angle
    return dict(padded_nodes=padded, time_steps=[0.0, step, step],
                keyframes=[(p1[0], p1[1], rot), (p2[0], p2[1], rot)], interp="cosine", offset=0.0)


def keyframe_pose(obstacle: Dict, time: float):
    """(x, y, rotation) of an obstacle at ``time``: the cyclic key-frame animation of ``obstacle.py:71-88`` (host-side
    twin of what the kernel evaluates per step; used by callers that need obstacle positions, e.g. the MPC feeders)."""
    steps, frames = obstacle["time_steps"], obstacle["keyframes"]
    tm = (time + obstacle["offset"]) % float(sum(steps))
    t = 0.0
    for i in range(len(frames)):
        t += steps[i]
        if t <= tm < t + steps[i + 1]:
            x = (tm - t) / steps[i + 1]
            alpha = (1.0 - math.cos(x * math.pi)) / 2.0 if obstacle["interp"] == "cosine" else x
            k0, k1 = frames[i], frames[(i + 1) % len(frames)]
            return tuple(k0[j] * (1.0 - alpha) + k1[j] * alpha for j in range(3))
    return tuple(frames[-1])


def make_map(boundary, static: Sequence, dynamic: Sequence[Dict], start, goal, path,
             radius: float = ROBOT["radius"]) -> Dict:
    """Map description -> the spec both the record packer and ``oracle/rl_env_numpy.py`` consume."""
    obstacles = [static_obstacle(n, radius) for n in static]
    obstacles += [periodic_obstacle(radius=radius, **d) for d in dynamic]
    return dict(start=np.asarray(start, dtype=np.float64), goal=_f32(goal)[:2],
                path=np.asarray(path, dtype=np.float64).reshape(-1, 2),
                boundary_padded=rg.buffer_polygon(_f32(rg.orient(boundary)), -radius), obstacles=obstacles)


class _CParams(C.Structure):
    _fields_ = [("n_path_max", C.c_int32), ("n_obst_max", C.c_int32), ("n_kf_max", C.c_int32), ("n_edge_max", C.c_int32),
                ("num_segments", C.c_int32), ("corner_samples", C.c_int32),
                ("time_step", C.c_double), ("sample_offset", C.c_double), ("collision_factor", C.c_double),
                ("reach_goal_factor", C.c_double), ("cross_track_factor", C.c_double),
                ("excessive_speed_factor", C.c_double), ("reference_speed", C.c_double),
                ("path_progress_factor", C.c_double),
                ("radius", C.c_double), ("speed_min", C.c_double), ("speed_max", C.c_double), ("angvel_min", C.c_double),
                ("angvel_max", C.c_double), ("acc_min", C.c_double), ("acc_max", C.c_double), ("angacc_min", C.c_double),
                ("angacc_max", C.c_double)]


ENV_EXPORTS = ("mpcgpu_env_record_doubles", "mpcgpu_env_step_dev", "mpcgpu_env_step_autoreset_dev", "mpcgpu_env_last_error")


def _bind(lib):
    if getattr(lib, "_env_bound", False):
        return lib
    vp = C.c_void_p
    lib.mpcgpu_env_record_doubles.argtypes = [C.POINTER(_CParams)]
    lib.mpcgpu_env_record_doubles.restype = C.c_int32
    lib.mpcgpu_env_step_dev.argtypes = [C.c_int32, C.POINTER(_CParams), C.c_int32] + [vp] * 8
    lib.mpcgpu_env_step_dev.restype = C.c_int32
    lib.mpcgpu_env_step_autoreset_dev.argtypes = [C.c_int32, C.POINTER(_CParams), C.c_int32] + [vp] * 10 + [C.c_int32, vp]
    lib.mpcgpu_env_step_autoreset_dev.restype = C.c_int32
    lib.mpcgpu_env_last_error.argtypes = []
    lib.mpcgpu_env_last_error.restype = C.c_char_p
    lib._env_bound = True
    return lib


def path_lengths(path: np.ndarray):
    """(cumulative length per node, length per segment), summed sequentially in float64 so that
    ``cum[i + 1] == cum[i] + seg[i]`` holds exactly (projection onto a node and the corner search must agree)."""
    n = len(path)
    cum, seg = np.zeros(n), np.zeros(n)
    for i in range(n - 1):
        seg[i] = math.sqrt((path[i + 1][0] - path[i][0]) ** 2 + (path[i + 1][1] - path[i][1]) ** 2)
        cum[i + 1] = cum[i] + seg[i]
    return cum, seg


def pack_records(maps: Sequence[Dict], n_kf_max: int = 2):
    """maps -> (records [B, R] float64, dict of the batch maxima P, M, K, E); layout: include/mpcgpu_env.h."""
    P = max(2, max(len(m["path"]) for m in maps))
    M = max(len(m["obstacles"]) for m in maps)
    K = max([n_kf_max] + [len(o["keyframes"]) for m in maps for o in m["obstacles"]])
    E = max(len(m["boundary_padded"]) + sum(len(o["padded_nodes"]) for o in m["obstacles"]) for m in maps)
    if P > 64 or M > 31 or K > 4:
        raise ValueError("limits: at most 64 path nodes, 31 obstacles, 4 key frames per obstacle")
    an = 4 + (K + 1) + 3 * K
    o_cum, o_len, o_xy = HDR, HDR + P, HDR + 2 * P
    o_anim = o_xy + 2 * P
    o_edge = o_anim + M * an
    R = o_edge + 5 * E
    R += R & 1
    rec = np.zeros((len(maps), R))
    for b, m in enumerate(maps):
        r = rec[b]
        path = np.asarray(m["path"], dtype=np.float64)
        cum, seg = path_lengths(path)
        n = len(path)
        r[0], r[1], r[3], r[4] = n, len(m["obstacles"]), m["goal"][0], m["goal"][1]
        r[5:10] = np.asarray(m["start"], dtype=np.float64)
        r[o_cum:o_cum + n], r[o_len:o_len + n] = cum, seg
        r[o_xy:o_xy + 2 * n] = path.reshape(-1)
        edges = []
        ring = np.asarray(m["boundary_padded"], dtype=np.float64)
        edges += [(*ring[i], *ring[(i + 1) % len(ring)], -1.0) for i in range(len(ring))]
        for j, ob in enumerate(m["obstacles"]):
            a = r[o_anim + j * an:o_anim + (j + 1) * an]
            nk = len(ob["keyframes"])
            a[0] = 1.0 if ob["interp"] == "cosine" else 0.0
            a[1], a[2], a[3] = ob["offset"], nk, float(sum(ob["time_steps"]))
            a[4:4 + nk + 1] = ob["time_steps"]
            a[4 + K + 1:4 + K + 1 + 3 * nk] = np.asarray(ob["keyframes"], dtype=np.float64).reshape(-1)
            ring = np.asarray(ob["padded_nodes"], dtype=np.float64)
            edges += [(*ring[i], *ring[(i + 1) % len(ring)], float(j)) for i in range(len(ring))]
        r[2] = len(edges)
        blk = np.full((E, 5), 0.0)
        blk[:, 4] = -2.0
        blk[:len(edges)] = np.asarray(edges)
        r[o_edge:o_edge + 5 * E] = blk.reshape(-1)
    return rec, dict(n_path_max=P, n_obst_max=M, n_kf_max=K, n_edge_max=E)


class BatchedRaysEnv:
    """B independent ``TrajectoryPlannerEnvironmentRaysReward1`` environments stepped by one kernel launch.

    ``maps``: one spec per environment (:func:`make_map`).  Observations / rewards / flags are torch tensors on the
    device.  ``max_episode_steps`` is the gym ``TimeLimit`` the reference registers (environment/__init__.py:15-25)."""

    def __init__(self, maps: Sequence[Dict], device: int = 0, time_step: float = 0.2, max_episode_steps: int = 1000,
                 sample_offset: float = 0.0, collision_factor: float = 4.0, reach_goal_factor: float = 3.0,
                 cross_track_factor: float = 0.05, reference_speed: float = ROBOT["speed_max"] * 0.8,
                 path_progress_factor: float = 2.0):
        import torch
        if not torch.cuda.is_available():
            raise MpcGpuError("BatchedRaysEnv needs a HIP device: the environment step is a GPU kernel, there is no CPU path")
        self._torch = torch
        self._lib = _bind(load_library())
        self.device = torch.device("cuda", device)
        self.device_index = device
        self.B = len(maps)
        rec, maxima = pack_records(maps)
        self.params = _CParams(num_segments=8, corner_samples=3, time_step=time_step, sample_offset=sample_offset,
                               collision_factor=collision_factor, reach_goal_factor=reach_goal_factor,
                               cross_track_factor=cross_track_factor, excessive_speed_factor=2.0 * path_progress_factor,
                               reference_speed=reference_speed, path_progress_factor=path_progress_factor,
                               **maxima, **ROBOT)
        R = self._lib.mpcgpu_env_record_doubles(C.byref(self.params))
        if R != rec.shape[1]:
            raise MpcGpuError(f"record layout mismatch: library {R} doubles, packer {rec.shape[1]} "
                              f"({self._lib.mpcgpu_env_last_error().decode()})")
        self.records = torch.from_numpy(rec).to(self.device)
        start = np.stack([np.asarray(m["start"], dtype=np.float64) for m in maps])
        self._start = torch.from_numpy(start).to(self.device)
        self.state = torch.zeros(self.B, STATE_DOUBLES, dtype=torch.float64, device=self.device)
        self.obs_internal = torch.zeros(self.B, N_INTERNAL, dtype=torch.float32, device=self.device)
        self.obs_external = torch.zeros(self.B, N_EXTERNAL, dtype=torch.float32, device=self.device)
        self.reward = torch.zeros(self.B, dtype=torch.float64, device=self.device)
        self.terminated = torch.zeros(self.B, dtype=torch.uint8, device=self.device)
        self.truncated = torch.zeros(self.B, dtype=torch.uint8, device=self.device)
        self.term_internal = torch.zeros_like(self.obs_internal)
        self.term_external = torch.zeros_like(self.obs_external)
        self.max_episode_steps = max_episode_steps
        self.time_step = time_step

    # ---- kernel launch ---------------------------------------------------------------------------------------------
    def _launch(self, actions) -> None:
        torch = self._torch
        stream = torch.cuda.current_stream(self.device).cuda_stream
        aptr = None
        if actions is not None:
            actions = torch.as_tensor(actions, device=self.device).to(torch.int32).contiguous()
            if actions.shape != (self.B,):
                raise ValueError(f"actions must have shape ({self.B},)")
            aptr = actions.data_ptr()
        rc = self._lib.mpcgpu_env_step_dev(self.device_index, C.byref(self.params), self.B, self.records.data_ptr(),
                                           self.state.data_ptr(), aptr, self.obs_internal.data_ptr(),
                                           self.obs_external.data_ptr(), self.reward.data_ptr(),
                                           self.terminated.data_ptr(), stream)
        if rc != 0:
            raise MpcGpuError(self._lib.mpcgpu_env_last_error().decode())

    def _obs(self) -> Dict[str, "object"]:
        return {"internal": self.obs_internal.clone(), "external": self.obs_external.clone()}

    # ---- gym-style API ---------------------------------------------------------------------------------------------
    def reset(self, mask=None):
        """Reset all (or the masked) environments to their map's start state (environment.py:166-186).  The
        observation memory is NOT cleared: the reference's component keeps ``old_obs`` across episodes."""
        torch = self._torch
        if mask is None:
            mask = torch.ones(self.B, dtype=torch.bool, device=self.device)
        mask = torch.as_tensor(mask, device=self.device).bool()
        keep = self.state[:, 8:24].clone()
        fresh = torch.zeros_like(self.state)
        fresh[:, :5] = self._start
        fresh[:, 8:24] = keep
        self.state = torch.where(mask[:, None], fresh, self.state)
        # update_status(reset=True) + get_observation for the reset rows only: the launch observes every environment, so the
        # rows that are NOT reset get their state, observation (including its one-step memory half), reward and
        # termination flag restored afterwards -- a partial reset must leave them bit-identical
        others = self.state.clone()
        kept = (self._obs(), self.reward.clone(), self.terminated.clone())
        self._launch(None)
        self.state = torch.where(mask[:, None], self.state, others)
        self.state[:, 6] = torch.where(mask, torch.zeros_like(self.state[:, 6]), self.state[:, 6])
        merged = self._merge(kept[0], self._obs(), mask)
        self.obs_internal.copy_(merged["internal"]); self.obs_external.copy_(merged["external"])
        self.reward.copy_(torch.where(mask, self.reward, kept[1]))
        self.terminated.copy_(torch.where(mask, self.terminated, kept[2]))
        return self._obs()

    def state_dict(self) -> Dict:
        """Everything a continuation needs: robot / clock state (incl. the one-step observation memory kept in it) and the
        observation, reward and flags of the last launch."""
        return dict(state=self.state.clone(), obs_internal=self.obs_internal.clone(), obs_external=self.obs_external.clone(),
                    reward=self.reward.clone(), terminated=self.terminated.clone(), truncated=self.truncated.clone())

    def load_state_dict(self, d: Dict) -> None:
        for k in ("state", "obs_internal", "obs_external", "reward", "terminated", "truncated"):
            getattr(self, k).copy_(d[k].to(self.device))

    def step(self, actions, auto_reset: bool = False):
        """``env.step`` (environment.py:199-213) -> (obs, reward [B], terminated [B] bool, truncated [B] bool, info).

        ``auto_reset=True`` is the vectorised-environment behaviour (SB3 VecEnv + gym TimeLimit): ended episodes are reset
        INSIDE the same kernel launch, ``obs`` is then the first observation of the next episode and
        ``info["terminal_observation"]`` the one the episode ended in (for the other rows it equals ``obs``)."""
        torch = self._torch
        if not auto_reset:
            self._launch(actions)
            obs = self._obs()
            terminated = self.terminated.bool()
            truncated = (self.state[:, 25] >= self.max_episode_steps) & ~terminated
            return obs, self.reward.clone(), terminated, truncated, {"success": (self.state[:, 7].to(torch.int64) & 4) != 0}
        stream = torch.cuda.current_stream(self.device).cuda_stream
        actions = torch.as_tensor(actions, device=self.device).to(torch.int32).contiguous()
        if actions.shape != (self.B,):
            raise ValueError(f"actions must have shape ({self.B},)")
        rc = self._lib.mpcgpu_env_step_autoreset_dev(
            self.device_index, C.byref(self.params), self.B, self.records.data_ptr(), self.state.data_ptr(),
            actions.data_ptr(), self.obs_internal.data_ptr(), self.obs_external.data_ptr(), self.reward.data_ptr(),
            self.terminated.data_ptr(), self.truncated.data_ptr(), self.term_internal.data_ptr(),
            self.term_external.data_ptr(), int(self.max_episode_steps), stream)
        if rc != 0:
            raise MpcGpuError(self._lib.mpcgpu_env_last_error().decode())
        obs = self._obs()
        terminated, truncated = self.terminated.bool(), self.truncated.bool()
        done = terminated | truncated
        # state[7] already belongs to the next episode where a reset happened; state[26] keeps this step's flags
        info = {"success": (self.state[:, 26].to(torch.int64) & 4) != 0,
                "terminal_observation": {"internal": torch.where(done[:, None], self.term_internal, obs["internal"]),
                                         "external": torch.where(done[:, None], self.term_external, obs["external"])}}
        return obs, self.reward.clone(), terminated, truncated, info

    def _merge(self, old, new, mask):
        torch = self._torch
        return {k: torch.where(mask[:, None], new[k], old[k]) for k in old}

    def set_agent_state(self, states) -> None:
        """``set_agent_state`` (environment.py:189-193) for every environment: rows of (x, y, theta, v, w)."""
        torch = self._torch
        self.state[:, :5] = torch.as_tensor(states, dtype=torch.float64, device=self.device)

    def observe(self):
        """``update_status(reset=False)`` + ``get_observation()`` without moving anything (src/main.py:181-189)."""
        self._launch(None)
        return self._obs()

    # ---- convenience -------------------------------------------------------------------------------------------------
    @property
    def agent_state(self):
        return self.state[:, :5]

    @property
    def path_progress(self):
        return self.state[:, 24]

    @property
    def flags(self):
        """[B, 3] bool: collided with obstacle, collided with boundary, reached goal."""
        f = self.state[:, 7].to(self._torch.int64)
        return self._torch.stack([(f & 1) != 0, (f & 2) != 0, (f & 4) != 0], dim=1)
