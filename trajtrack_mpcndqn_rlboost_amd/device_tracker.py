"""Batched tracker with DEVICE-RESIDENT state: the tick path of :class:`batched_tracker.BatchedTracker` without the padded
parameter vectors and without a host round trip.

Per tick (``step``): termination test, speed rule and assembly written directly as the solver's compact workspace record
(``tracker_assemble_kernel`` reads the indices of the reference's parameter vector, ``mpc_generator.py:179-188``, straight from
the arrays below -- bitwise the record the compaction kernel makes of ``BatchedTracker.assemble()``), the batched solve, and the
post-solve rollouts (``trajectory_generator.py:325-339``) -- all enqueued on one stream, nothing read back.  ``local_refs`` is the
window search of ``get_local_ref_traj`` (``trajectory_generator.py:206-232``) as a kernel.  Set-up calls (``initialization``,
``update_static_constraints``) run on the host exactly like the host tracker's and are uploaded before the next tick -- per
robot and per field: ``update_static_constraints(i)`` rewrites robot i's half-plane rows and nothing else (it may be called at any
time, ``src/interface_mpc.py:60-63``), ``initialization(i)`` re-plans robot i alone; the other robots keep the state the device has
advanced them to.
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np

from .config import MpcConfig
from .geometry import static_obstacle_params
from .solver import BatchSolver, CTracker
from .trajectory_generator import global_reference_trajectory, work_mode


class DeviceTracker:
    def __init__(self, config: MpcConfig, n_robots: int, device: int = 0, solver: Optional[BatchSolver] = None,
                 mode: str = "work"):
        import torch
        self._torch = torch
        self.config, self.B = config, int(n_robots)
        self.device = torch.device("cuda", device)
        self.solver = solver if solver is not None else BatchSolver(config, device=device)
        N, B = config.N_hor, self.B
        f64 = dict(dtype=torch.float64, device=self.device)
        self.states = torch.zeros(B, config.ns, **f64)
        self.goals = torch.zeros(B, config.ns, **f64)
        self.last_actions = torch.zeros(B, config.nu, **f64)
        self.idx_ref = torch.zeros(B, dtype=torch.int32, device=self.device)
        self.ref_len = torch.ones(B, dtype=torch.int32, device=self.device)
        self.ref = torch.zeros(B, 1, 3, **f64)
        self.stc = torch.zeros(B, config.Nstcobs * config.nstcobs, **f64)
        self.dyn = torch.zeros(B, config.Ndynobs * config.ndynobs * N, **f64)
        self.other = torch.zeros(B, config.ns * N * config.Nother, **f64)
        self.pred_states = torch.zeros(B, N, config.ns, **f64)
        self.active = torch.ones(B, dtype=torch.uint8, device=self.device)
        self.refs = torch.zeros(B, N, 3, **f64)            # the window of the last local_refs() call
        self.out = dict(u=torch.zeros(B, 2 * N, **f64), cost=torch.zeros(B, **f64),
                        status=torch.zeros(B, dtype=torch.int32, device=self.device),
                        inner_it=torch.zeros(B, dtype=torch.int32, device=self.device),
                        outer_it=torch.zeros(B, dtype=torch.int32, device=self.device), actions=torch.zeros(B, 2, **f64))
        self.stc_weight = self.dyn_weight = 1e3
        self.set_mode(mode)
        # host mirrors of what the set-up calls fill; uploaded before the next tick
        self._h_ref = [np.zeros((1, 3))] * B
        self._h_states, self._h_goals = np.zeros((B, 3)), np.zeros((B, 3))
        self._h_stc = np.zeros((B, config.Nstcobs * config.nstcobs))
        self._init_rows, self._stc_rows = set(), set()      # robots whose set-up calls wait for their upload

    def set_mode(self, mode: str):
        self.base_speed, self.tuning = work_mode(self.config, mode)

    # -- per-robot set-up (host, like BatchedTracker's) ----------------------------------------------------------------
    def initialization(self, i: int, init_state, goal_state, ref_path_list: Sequence[Sequence[float]], mode: str = "work"):
        base_speed, _ = work_mode(self.config, mode)
        self._h_states[i], self._h_goals[i] = init_state, goal_state
        self._h_ref[i] = global_reference_trajectory(self.config.ts, ref_path_list, self._h_states[i], base_speed)
        self._init_rows.add(int(i))

    def update_static_constraints(self, i: int, obstacle_list):
        self._h_stc[i] = static_obstacle_params(obstacle_list, self.config.Nstcobs, self.config.nstcobs)
        self._stc_rows.add(int(i))

    def _upload(self):
        """Rows touched by set-up calls since the last tick, and only those: the device arrays are the truth for every robot
        that is running (the host mirrors hold what the set-up calls were given, not where the robots are now)."""
        torch = self._torch
        if self._stc_rows:
            rows = sorted(self._stc_rows)
            self.stc[torch.tensor(rows, device=self.device)] = torch.from_numpy(self._h_stc[rows]).to(self.device)
            self._stc_rows.clear()
        if self._init_rows:
            rows = sorted(self._init_rows)
            idx = torch.tensor(rows, device=self.device)
            need = max(len(self._h_ref[i]) for i in rows)
            if need > self.ref.shape[1]:        # longer reference than any so far: grow the table, the others keep their rows
                grown = torch.zeros(self.B, need, 3, dtype=torch.float64, device=self.device)
                grown[:, :self.ref.shape[1]] = self.ref
                self.ref = grown
            block = np.zeros((len(rows), self.ref.shape[1], 3))
            for j, i in enumerate(rows):
                block[j, :len(self._h_ref[i])] = self._h_ref[i]
            self.ref[idx] = torch.from_numpy(block).to(self.device)
            self.ref_len[idx] = torch.tensor([len(self._h_ref[i]) for i in rows], dtype=torch.int32, device=self.device)
            self.states[idx] = torch.from_numpy(self._h_states[rows]).to(self.device)
            self.goals[idx] = torch.from_numpy(self._h_goals[rows]).to(self.device)
            self.last_actions[idx] = 0.0
            self.idx_ref[idx] = 0
            self.active[idx] = 1
            self._init_rows.clear()

    def set_dynamic_constraints(self, predictions):
        """``predictions`` [B, K, N, 6] (tensor on this device or array): rows of the first K dynamic-obstacle slots."""
        torch = self._torch
        block = torch.as_tensor(predictions, dtype=torch.float64, device=self.device).reshape(self.B, -1)
        self.dyn[:, :block.shape[1]] = block

    def set_other_robot_states(self, other):
        torch = self._torch
        self.other.copy_(torch.as_tensor(other, dtype=torch.float64, device=self.device).reshape(self.B, -1))

    # -- device view ---------------------------------------------------------------------------------------------------
    def view(self) -> CTracker:
        self._upload()
        v = CTracker()
        v.B, v.ref_cap, v.action_steps = self.B, int(self.ref.shape[1]), int(self.config.action_steps)
        for name, t in (("states", self.states), ("goals", self.goals), ("last_actions", self.last_actions), ("ref", self.ref),
                        ("ref_len", self.ref_len), ("idx_ref", self.idx_ref), ("stc", self.stc), ("dyn", self.dyn),
                        ("other", self.other), ("pred_states", self.pred_states), ("active", self.active)):
            assert t.is_contiguous()
            setattr(v, name, t.data_ptr())
        for j, w in enumerate(self.tuning):
            v.tuning[j] = float(w)
        v.base_speed, v.low_speed = float(self.base_speed), float(self.config.low_speed)
        v.stc_weight, v.dyn_weight = float(self.stc_weight), float(self.dyn_weight)
        return v

    # -- one control tick ------------------------------------------------------------------------------------------------
    def local_refs(self, stream: Optional[int] = None):
        """[B, N, 3] device tensor: every robot's local reference window; advances the indices (a view of the tracker's own
        buffer: valid until the next call)."""
        self.solver.tracker_window(self.view(), self.refs, stream=self._stream(stream))
        return self.refs

    def step(self, refs=None, initial_guess=None, stream: Optional[int] = None):
        """Enqueue the tick of all robots.  ``refs`` [B, N, 3] device tensor: the reference every robot tracks (default: its
        local window).  Returns the dict of device tensors ``u, cost, status, inner_it, outer_it, actions`` (the tracker's own
        buffers); ``states``, ``pred_states``, ``last_actions`` and ``active`` are updated in place."""
        if refs is None:
            refs = self.local_refs(stream)
        self.solver.tracker_step(self.view(), refs, self.out, initial_guess=initial_guess, stream=self._stream(stream))
        return self.out

    def _stream(self, stream):
        return self._torch.cuda.current_stream().cuda_stream if stream is None else stream
