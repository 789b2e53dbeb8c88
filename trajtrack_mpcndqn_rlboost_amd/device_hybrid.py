"""The hybrid decision loop of :class:`hybrid.BatchedHybrid` with the whole tick on the device (modes 1 and 2 of
``src/main.py:96-98``): obstacle positions and constant-velocity predictions (``main.py:77-85,127``) as tensor operations, the
DQN's proposal rollout and the HintSwitcher as kernels (``csrc/trackgpu.hip``), the tracker tick as
:class:`device_tracker.DeviceTracker` (assembly written directly as the solver's compact record, solve, rollouts).  The host
touches a tick once: a 3-byte-per-robot flag read-back at its end (collision / success / done bookkeeping).

Same constructor, same ``tick`` / ``run`` results as the host loop; ``tests/test_gpu_hybrid.py`` runs the two side by side."""
from __future__ import annotations

import time
from typing import Dict, Optional, Sequence

import numpy as np

from . import dqn as dqn_mod
from .config import MpcConfig
from .device_tracker import DeviceTracker
from .dqn import QNetwork
from .feeders import DYN_OBS_SIZE
from .hybrid import BatchedHybrid


class DeviceHybrid(BatchedHybrid):
    def __init__(self, config: MpcConfig, scenes: Sequence[Dict], q_net: QNetwork, decision_mode: int = 2, device: int = 0,
                 inflate_margin: float = 0.8, switcher=(10, 2, 10)):
        if decision_mode == 0:
            raise ValueError("decision_mode 0 (pure DQN) has no MPC tick: use BatchedHybrid")
        self._switch_params = tuple(float(v) for v in switcher)
        super().__init__(config, scenes, q_net, decision_mode, device, inflate_margin, switcher,
                         tracker=_HostShim(config, len(scenes)))
        torch = self._torch
        dev = self.env.device
        self.dtracker = DeviceTracker(config, self.B, device=device)
        # dynamic obstacles: two-key-frame cosine animations (rl_env.periodic_obstacle) as tensors [B, K]
        K = int(self._n_dynamic.max()) if self.B else 0
        p1, p2, step, offs = (np.zeros((self.B, K, 2)), np.zeros((self.B, K, 2)), np.ones((self.B, K)), np.zeros((self.B, K)))
        for b, m in enumerate(self.maps):
            for j, ob in enumerate(m["obstacles"][self._n_static[b]:]):
                if len(ob["keyframes"]) != 2 or ob["interp"] != "cosine" or ob["time_steps"][1] != ob["time_steps"][2]:
                    raise NotImplementedError("DeviceHybrid animates periodic two-key-frame obstacles (rl_env.periodic_obstacle)")
                p1[b, j], p2[b, j] = ob["keyframes"][0][:2], ob["keyframes"][1][:2]
                step[b, j], offs[b, j] = ob["time_steps"][1], ob["offset"]
        f64 = dict(dtype=torch.float64, device=dev)
        self._kf = tuple(torch.tensor(a, **f64) for a in (p1, p2, step, offs))
        self._has_dyn = torch.tensor(np.arange(K)[None, :] < self._n_dynamic[:, None], device=dev)
        self._K = K
        self._d_polygons = torch.tensor(self._polygons, **f64)
        self._d_valid = torch.tensor(self._poly_valid.astype(np.uint8), device=dev)
        self._corners = torch.tensor([[-DYN_OBS_SIZE, -DYN_OBS_SIZE], [DYN_OBS_SIZE, -DYN_OBS_SIZE], [DYN_OBS_SIZE, DYN_OBS_SIZE],
                                      [-DYN_OBS_SIZE, DYN_OBS_SIZE]], **f64)
        self._ksteps = torch.arange(1, config.N_hor + 1, **f64)
        self._rl_ref = torch.zeros(self.B, 20, 2, **f64)
        self._chosen = torch.zeros(self.B, config.N_hor, 3, **f64)
        self._limits = (dqn_mod.ACCELERATION_MAX, dqn_mod.ACCELERATION_MIN, dqn_mod.ANGULAR_ACCELERATION_MAX,
                        dqn_mod.ANGULAR_ACCELERATION_MIN, dqn_mod.SPEED_MIN, dqn_mod.SPEED_MAX, dqn_mod.ANGULAR_VELOCITY_MIN,
                        dqn_mod.ANGULAR_VELOCITY_MAX)
        self.reset()

    # the base class's reset() drives its tracker through the host API: mirrored onto the device tracker here
    def reset(self):
        if not hasattr(self, "dtracker"):
            return                      # called from the base constructor, before the device tracker exists
        torch = self._torch
        dev = self.env.device
        self.obs = self.env.reset()
        for i, s in enumerate(self.scenes):
            start = np.asarray(s["start"], dtype=float)
            self.dtracker.initialization(i, start[:3], np.array([s["goal"][0], s["goal"][1], 0.0]), s["path"])
            self.dtracker.update_static_constraints(i, self.inflated[i])
        self.dtracker.view()
        self.tracker.ref_trajs = [np.array(r) for r in self.dtracker._h_ref]
        self._sw_on = torch.zeros(self.B, dtype=torch.uint8, device=dev)
        self._sw_cnt = torch.zeros(self.B, dtype=torch.int32, device=dev)
        self._last_dyn = None
        self._done = torch.zeros(self.B, dtype=torch.bool, device=dev)
        self.done = np.zeros(self.B, dtype=bool)
        self.success = np.zeros(self.B, dtype=bool)
        self.collided = np.zeros(self.B, dtype=bool)
        self.steps = np.zeros(self.B, dtype=int)
        self.switch_on = np.zeros(self.B, dtype=bool)
        self.switch_ticks = np.zeros(self.B, dtype=int)
        self._d_switch_ticks = torch.zeros(self.B, dtype=torch.int64, device=dev)
        self.t = 0

    def dynamic_positions_device(self):
        """[B, K, 2] current key-frame positions (rl_env.keyframe_pose for every obstacle, as tensor operations)."""
        torch = self._torch
        p1, p2, step, offs = self._kf
        tm = torch.remainder(self.env.state[:, 5:6] + offs, 2.0 * step)
        first = tm < step
        x = torch.where(first, tm / step, (tm - step) / step)
        alpha = ((1.0 - torch.cos(x * np.pi)) / 2.0)[..., None]
        a = torch.where(first[..., None], p1, p2)
        b = torch.where(first[..., None], p2, p1)
        return a * (1.0 - alpha) + b * alpha

    def tick(self) -> Dict[str, np.ndarray]:
        torch, cfg, env, trk = self._torch, self.config, self.env, self.dtracker
        stream = torch.cuda.current_stream().cuda_stream
        self._mark("start")
        K, N = self._K, cfg.N_hor
        dyn_now = self.dynamic_positions_device()
        if self._last_dyn is None:
            self._last_dyn = dyn_now
        if K:                                                # est_dyn_obs_positions (main.py:77-85) for every robot
            delta = dyn_now - self._last_dyn
            preds = torch.zeros(self.B, K, N, 6, dtype=torch.float64, device=env.device)
            preds[..., 0] = dyn_now[:, :, None, 0] + delta[:, :, None, 0] * self._ksteps
            preds[..., 1] = dyn_now[:, :, None, 1] + delta[:, :, None, 1] * self._ksteps
            preds[..., 2] = DYN_OBS_SIZE
            preds[..., 3] = DYN_OBS_SIZE
            preds[..., 5] = 1.0
            preds = torch.where(self._has_dyn[:, :, None, None], preds, torch.zeros_like(preds))
        self._last_dyn = dyn_now
        live = ~self._done
        self._mark("obstacle predictions (device)")
        # the environment follows the tracker: position / heading from the MPC state, speeds from its last action
        env.state[:, :3] = trk.states
        env.state[:, 3:5] = trk.last_actions
        if self.mode == 1:                                   # main.py:154-157
            self.obs, _, term, trunc, info = env.step(torch.zeros(self.B, dtype=torch.int32))
            self._mark("environment kernel")
            chosen = trk.local_refs(stream)
            self._mark("local reference (device)")
        else:                                                # main.py:176-214
            actions = self.q_net.greedy_actions(self._flat(self.obs))
            self._mark("Q-network")
            env.state[:, 5] += env.time_step                 # step_obstacles()
            self.obs = env.observe()                         # update_status() + get_observation()
            term = env.terminated.bool()
            self._mark("environment kernel")
            trk.solver.rl_reference(env.state, actions, cfg.ts, 20, 1.0, self._limits, self._rl_ref, stream=stream)
            original = trk.local_refs(stream)
            if K:                                            # circle_to_rect of every disc (main.py:129)
                s0 = int(self._n_static.max())
                rects = dyn_now[:, :, None, :] + self._corners[None, None]
                self._d_polygons[:, s0:s0 + K, :4] = rects
                self._d_polygons[:, s0:s0 + K, 4:] = rects[:, :, 3:4]
            live_u8 = live.to(torch.uint8)
            trk.solver.hint_switch(self._d_polygons, self._d_valid, trk.states, original, self._rl_ref, live_u8,
                                   self._switch_params, self._sw_on, self._sw_cnt, self._chosen, stream=stream)
            chosen = self._chosen
            self._d_switch_ticks += (self._sw_on.bool() & live)
            self._mark("RL reference + switch (device)")
        if K:
            trk.set_dynamic_constraints(preds)
        trk.active &= live.to(torch.uint8)
        trk.step(refs=chosen, stream=stream)
        if getattr(self, "profile", False):
            torch.cuda.synchronize()
            t = trk.solver.last_timing()
            self.phase_seconds["  of which: assembly kernel"] = self.phase_seconds.get("  of which: assembly kernel", 0.0) + 1e-3 * t["prep_ms"]
            self.phase_seconds["  of which: solve kernel"] = self.phase_seconds.get("  of which: solve kernel", 0.0) + 1e-3 * t["solve_ms"]
        self._mark("assembly + batched MPC solve + rollouts (device)")
        # bookkeeping: ONE small read-back per tick
        flags = env.flags
        packed = torch.stack([flags[:, 0] | flags[:, 1], flags[:, 2], term.bool() | (live & ~trk.active.bool()),
                              self._sw_on.bool() & live], dim=1).cpu().numpy()
        live_h = ~self.done
        self.collided |= live_h & packed[:, 0]
        self.success |= live_h & packed[:, 1]
        self.done |= live_h & packed[:, 2]
        self._done = torch.from_numpy(self.done).to(env.device)
        self.switch_on = packed[:, 3]
        self.switch_ticks += self.switch_on
        self.steps += live_h
        self.t += 1
        self._mark("bookkeeping (flag read-back)")
        return dict(done=self.done.copy(), success=self.success.copy(), collided=self.collided.copy(),
                    switch_on=self.switch_on.copy(), states=trk.states.cpu().numpy())

    @property
    def tracker_states(self):
        return self.dtracker.states


class _HostShim:
    """What BatchedHybrid's constructor expects of a tracker, without a solver or host buffers behind it."""

    def __init__(self, config, B):
        self.config, self.B = config, B
        self.states = np.zeros((B, 3))

    def initialization(self, *a, **k):
        pass

    def update_static_constraints(self, *a, **k):
        pass

    ref_trajs = ()      # set by DeviceHybrid.reset(): the global reference trajectories (host copies kept by the device tracker)
