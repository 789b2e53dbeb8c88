"""Batched tracker: B robots, one GPU solve per control tick.

The single-robot harness of the reference (``src/mpc_traj_tracker/trajectory_generator.py:235-339``,
``src/interface_mpc.py:16-88``) is applied to every robot, but the parameter vectors are stacked into one
``[B, np]`` array and solved by ONE call of the batched C-ABI entry point; the post-solve rollouts are vectorised.
Robots are independent within a tick.  Fleet coupling comes in two forms: ``step()`` uses the previous tick's
predictions for every robot (Jacobi: ONE solve per tick); ``step(groups=...)`` reproduces the reference's sequential
loop (``src/scenario_simulator.py:226-233``: robot j sees the FRESH prediction of every robot i < j of its world --
Gauss-Seidel) colour by colour: the robots at position c of every group are independent of each other, so a tick of
G worlds x R robots is R batched solves of G problems.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np

from .config import MpcConfig
from .geometry import static_obstacle_params
from .motion_model import unicycle_model
from .solver import BatchSolver, BatchResult
from .trajectory_generator import global_reference_trajectory, work_mode


class BatchedTracker:
    """State of B robots as arrays; every per-tick step (window search, speed rule, parameter assembly, rollouts) is a
    numpy array operation over the batch -- no Python loop over robots on the tick path."""

    def __init__(self, config: MpcConfig, n_robots: int, device: int = 0, solver: Optional[BatchSolver] = None,
                 warm_start: bool = False):
        self.config = config
        self.B = int(n_robots)
        self.warm_start = warm_start        # default False = the reference's call sites (initial_guess=None -> u0 = 0)
        self.solver = solver if solver is not None else BatchSolver(config, device=device)
        N = config.N_hor
        self.states = np.zeros((self.B, config.ns))
        self.goals = np.zeros((self.B, config.ns))
        self.last_actions = np.zeros((self.B, config.nu))
        self._ref = np.zeros((self.B, 1, 3))            # global reference trajectories, padded to the longest
        self._ref_len = np.ones(self.B, dtype=np.int64)
        self.idx_ref = np.zeros(self.B, dtype=np.int64)
        self.stc_constraints = np.zeros((self.B, config.Nstcobs * config.nstcobs))
        self.dyn_constraints = np.zeros((self.B, config.Ndynobs * config.ndynobs * N))
        self.other_robot_states = np.zeros((self.B, config.ns * N * config.Nother))
        self.stc_weights = [1e3] * N
        self.dyn_weights = [1e3] * N
        self.pred_states = np.zeros((self.B, N, config.ns))
        self.active = np.ones(self.B, dtype=bool)   # False once a robot's termination test has fired
        # InterfaceMpc.get_action stops serving a robot whose termination test has fired (src/interface_mpc.py:83-85); the
        # multi-robot simulator keeps calling run_step for every robot until ALL have arrived (src/scenario_simulator.py:226-250).
        # False reproduces the latter: the test is still evaluated (``arrived``), but nobody is frozen.
        self.stop_when_done = True
        self.arrived = np.zeros(self.B, dtype=bool)
        self.last_result: Optional[BatchResult] = None
        self._P: Optional[np.ndarray] = None

    @property
    def ref_trajs(self) -> List[np.ndarray]:
        return [self._ref[i, :self._ref_len[i]] for i in range(self.B)]

    # -- per-robot set-up (same meaning as InterfaceMpc.initialization / update_*_constraints) ---------------------
    def initialization(self, i: int, init_state, goal_state, ref_path_list: Sequence[Sequence[float]], mode: str = "work"):
        self.states[i] = init_state
        self.goals[i] = goal_state
        self.last_actions[i] = 0.0
        base_speed, _ = work_mode(self.config, mode)
        traj = global_reference_trajectory(self.config.ts, ref_path_list, self.states[i], base_speed)
        if len(traj) > self._ref.shape[1]:
            grown = np.zeros((self.B, len(traj), 3))
            grown[:, :self._ref.shape[1]] = self._ref
            self._ref = grown
        self._ref[i, :len(traj)] = traj
        self._ref_len[i] = len(traj)
        self.idx_ref[i] = 0
        self.active[i] = True

    def update_static_constraints(self, i: int, obstacle_list):
        self.stc_constraints[i] = static_obstacle_params(obstacle_list, self.config.Nstcobs, self.config.nstcobs)

    def update_dynamic_constraints(self, i: int, full_dyn_obstacle_list):
        """``full_dyn_obstacle_list[j][k]`` = (x, y, rx, ry, angle, alpha) of obstacle j at step k; slots beyond the
        list keep what they held (``interface_mpc.py:65-67``)."""
        block = np.asarray(full_dyn_obstacle_list, dtype=float).reshape(-1)
        self.dyn_constraints[i, :block.size] = block

    def set_dynamic_constraints(self, predictions: np.ndarray):
        """All robots at once: ``predictions`` [B, K, N, 6] (e.g. ``feeders.constant_velocity_prediction``)."""
        block = np.asarray(predictions, dtype=float).reshape(self.B, -1)
        self.dyn_constraints[:, :block.shape[1]] = block

    def share_predictions(self, groups: Optional[Sequence[Sequence[int]]] = None):
        """Fill every robot's other-robot block with the latest predictions of the other robots of its group, in group
        order, N x (x, y, theta) each, zero padded -- ``get_other_robot_states`` (``scenario_simulator.py:154-163``).
        Before the first solve the predictions are zeros, which is what the reference's block holds then too."""
        cfg = self.config
        groups = groups if groups is not None else [list(range(self.B))]
        per = cfg.N_hor * cfg.ns
        for g in groups:
            idx = np.asarray(g, dtype=np.int64)
            n = len(idx)
            if n == 0:
                continue
            keep = min(n - 1, cfg.Nother)
            block = np.zeros((n, cfg.Nother * per))
            if keep > 0:
                # others[i] = the group without robot i, order preserved
                others = np.stack([np.delete(idx, i)[:keep] for i in range(n)]) if n <= 64 else \
                    np.array([np.delete(idx, i)[:keep] for i in range(n)])
                block[:, :keep * per] = self.pred_states[others].reshape(n, keep * per)
            self.other_robot_states[idx] = block

    # -- one control tick for all robots ---------------------------------------------------------------------------
    def local_refs(self) -> np.ndarray:
        """``get_local_ref_traj`` for every robot ([B, N, 3]): nearest sample inside the window [idx-1, idx+5) action
        steps (first one on ties), then N rows from there, the tail padded with the last sample; advances the
        indices like the reference's call does (``trajectory_generator.py:206-232``, ``interface_mpc.py:73-80``).
        Pass a (modified) copy to :meth:`step` as ``refs``."""
        cfg = self.config
        a, rows = cfg.action_steps, np.arange(self.B)
        lb = np.maximum(0, self.idx_ref - a)
        ub = np.minimum(self._ref_len, self.idx_ref + 5 * a)
        cand = lb[:, None] + np.arange(6 * a)[None, :]
        valid = cand < ub[:, None]
        pts = self._ref[rows[:, None], np.minimum(cand, self._ref_len[:, None] - 1)]
        d = np.hypot(self.states[:, None, 0] - pts[..., 0], self.states[:, None, 1] - pts[..., 1])
        d[~valid] = np.inf
        self.idx_ref = lb + np.argmin(d, axis=1)
        take = np.minimum(self.idx_ref[:, None] + np.arange(cfg.N_hor)[None, :], self._ref_len[:, None] - 1)
        return self._ref[rows[:, None], take]

    def speed_refs(self, base_speed: float) -> np.ndarray:
        """[B] speed reference of this tick (``trajectory_generator.py:257-264``)."""
        cfg = self.config
        dist = np.hypot(self.states[:, 0] - self.goals[:, 0], self.states[:, 1] - self.goals[:, 1])
        return np.where(dist >= base_speed * cfg.N_hor * cfg.ts, base_speed,
                        np.maximum(dist / cfg.N_hor / cfg.ts, cfg.low_speed))

    def assemble(self, mode: str = "work", refs: Optional[np.ndarray] = None) -> np.ndarray:
        """[B, np] parameter vectors in the order of ``mpc_generator.py:179-188`` (a view of the tracker's own buffer:
        valid until the next call)."""
        cfg = self.config
        base_speed, tuning = work_mode(cfg, mode)
        if refs is None:
            refs = self.local_refs()
        refs = np.asarray(refs, dtype=float)
        off, N = cfg.offsets(), cfg.N_hor
        if self._P is None:                 # one buffer for the tracker's lifetime: every block is rewritten each tick
            self._P = np.zeros((self.B, cfg.num_params))
        P = self._P
        P[:, 0:3] = self.states
        P[:, 3:6] = refs[:, -1]
        P[:, 6:8] = self.last_actions
        P[:, off["q"]:off["q"] + len(tuning)] = tuning
        P[:, off["r"]:off["r"] + 3 * N] = refs.reshape(self.B, -1)
        P[:, off["vref"]:off["vref"] + N] = self.speed_refs(base_speed)[:, None]
        P[:, off["c"]:off["c"] + self.other_robot_states.shape[1]] = self.other_robot_states
        P[:, off["os"]:off["os"] + self.stc_constraints.shape[1]] = self.stc_constraints
        P[:, off["od"]:off["od"] + self.dyn_constraints.shape[1]] = self.dyn_constraints
        P[:, off["qstc"]:off["qstc"] + N] = self.stc_weights
        P[:, off["qdyn"]:off["qdyn"] + N] = self.dyn_weights
        return P

    def step(self, mode: str = "work", initial_guess: Optional[np.ndarray] = None, refs: Optional[np.ndarray] = None,
             groups: Optional[Sequence[Sequence[int]]] = None):
        """Solve all robots, apply the first ``action_steps`` inputs.  Returns (actions [B, nu], pred_states
        [B, N, ns], cost [B]); robots that already reached their goal keep their state (action 0).
        ``refs`` [B, N, 3]: the reference each robot tracks this tick (``get_action(current_ref_traj)``,
        ``interface_mpc.py:82-88``); default: the local window of its global reference.
        ``groups``: lists of mutually coupled robots (the robots of one world, in the reference's dictionary order).
        Given, the tick is solved Gauss-Seidel like the reference's loop (``scenario_simulator.py:226-233``): colour c =
        the c-th robot of every group; before colour c is solved its other-robot blocks are refreshed with the
        predictions colours < c have just produced (and last tick's for colours > c).  Not given: one solve for the
        whole batch with whatever ``other_robot_states`` holds (Jacobi when fed by ``share_predictions``)."""
        cfg = self.config
        near = np.all(np.abs(self.states[:, :2] - self.goals[:, :2]) <= 0.05, axis=1)
        self.arrived = near & (np.abs(self.last_actions[:, 0]) < 0.05)        # check_termination_condition
        if self.stop_when_done:
            self.active &= ~self.arrived
        if initial_guess is None and self.warm_start and self.last_result is not None:
            # receding-horizon warm start (what OpEn's TCP server does with its cached solution): previous plan
            # shifted by the inputs already applied, last input repeated
            prev = self.last_result.solution.reshape(self.B, cfg.N_hor, cfg.nu)
            k = cfg.action_steps
            initial_guess = np.concatenate([prev[:, k:], np.repeat(prev[:, -1:], k, axis=1)], axis=1).reshape(self.B, -1)
        if groups is None:
            P = self.assemble(mode, refs)
            res = self.solver.solve(P, initial_guess)
            self.last_result = res
            actions = self._apply(np.arange(self.B), res.solution)
            return actions, self.pred_states.copy(), res.cost
        # ---- Gauss-Seidel over colours
        if refs is None:
            refs = self.local_refs()        # a robot's window depends on its own state only: taken once, up front
        covered = np.sort(np.concatenate([np.asarray(g, dtype=np.int64) for g in groups])) if len(groups) else np.array([])
        if not np.array_equal(covered, np.arange(self.B)):
            raise ValueError("groups must partition the robots 0..B-1")
        actions = np.zeros((self.B, cfg.nu))
        parts = []
        for c in range(max(len(g) for g in groups)):
            idx = np.array([g[c] for g in groups if len(g) > c], dtype=np.int64)
            self.share_predictions(groups)
            P = self.assemble(mode, refs)
            res = self.solver.solve(P[idx], None if initial_guess is None else initial_guess[idx])
            actions[idx] = self._apply(idx, res.solution)
            parts.append((idx, res))
        self.last_result = _merge_results(self.B, parts)
        return actions, self.pred_states.copy(), self.last_result.cost

    def _apply(self, idx: np.ndarray, solution: np.ndarray) -> np.ndarray:
        """Post-solve part of ``run_step`` for the robots ``idx`` (``trajectory_generator.py:325-339``): advance by the
        first ``action_steps`` inputs, roll the prediction out; returns their applied first actions."""
        cfg = self.config
        u = solution.reshape(len(idx), cfg.N_hor, cfg.nu)
        state = self.states[idx].copy()
        for s in range(cfg.action_steps):
            state = unicycle_model(state, u[:, s], cfg.ts)
        pred = np.empty((len(idx), cfg.N_hor, cfg.ns))
        rolling = state
        for k in range(cfg.N_hor):                      # rolled from the taken state, re-applying u[0] (reference quirk)
            rolling = unicycle_model(rolling, u[:, k], cfg.ts)
            pred[:, k] = rolling
        act = self.active[idx]
        on, off = idx[act], idx[~act]
        self.states[on] = state[act]
        self.last_actions[on] = u[act, cfg.action_steps - 1]
        self.last_actions[off] = 0.0
        self.pred_states[on] = pred[act]
        return np.where(act[:, None], u[:, 0], 0.0)


def _merge_results(B: int, parts) -> BatchResult:
    """One BatchResult for the whole fleet from the per-colour results."""
    first = parts[0][1]
    fields = {}
    for name in ("solution", "cost", "status", "num_inner_iterations", "num_outer_iterations", "last_problem_norm_fpr",
                 "f2_norm", "lagrange_multipliers", "solve_time_ms"):
        proto = getattr(first, name)
        full = np.zeros((B,) + proto.shape[1:], dtype=proto.dtype)
        for idx, res in parts:
            full[idx] = getattr(res, name)
        fields[name] = full
    return BatchResult(**fields)
