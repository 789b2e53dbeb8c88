"""Batched tracker: B robots, one GPU solve per control tick.

The single-robot harness of the reference (``src/mpc_traj_tracker/trajectory_generator.py:235-339``,
``src/interface_mpc.py:16-88``) is applied to every robot, but the parameter vectors are stacked into one
``[B, np]`` array and solved by ONE call of the batched C-ABI entry point; the post-solve rollouts are vectorised.
Robots are independent within a tick.  Fleet coupling uses the previous tick's predictions for every robot
(Jacobi), whereas the reference's sequential loop (``src/scenario_simulator.py:226-233``) lets robot j see robot
i < j's fresh prediction (Gauss-Seidel) -- a documented difference for B > 1 fleets.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np

from .config import MpcConfig
from .geometry import static_obstacle_params
from .motion_model import unicycle_model
from .solver import BatchSolver, BatchResult
from .trajectory_generator import (assemble_parameters, global_reference_trajectory, local_reference_window,
                                   speed_references, work_mode)


class BatchedTracker:
    def __init__(self, config: MpcConfig, n_robots: int, device: int = 0, solver: Optional[BatchSolver] = None):
        self.config = config
        self.B = int(n_robots)
        self.solver = solver if solver is not None else BatchSolver(config, device=device)
        N = config.N_hor
        self.states = np.zeros((self.B, config.ns))
        self.goals = np.zeros((self.B, config.ns))
        self.last_actions = np.zeros((self.B, config.nu))
        self.ref_trajs: List[np.ndarray] = [np.zeros((1, 3))] * self.B
        self.idx_ref = [0] * self.B
        self.stc_constraints = [[0.0] * (config.Nstcobs * config.nstcobs) for _ in range(self.B)]
        self.dyn_constraints = [[0.0] * (config.Ndynobs * config.ndynobs * N) for _ in range(self.B)]
        self.other_robot_states = [[0.0] * (config.ns * N * config.Nother) for _ in range(self.B)]
        self.stc_weights = [1e3] * N
        self.dyn_weights = [1e3] * N
        self.pred_states = np.zeros((self.B, N, config.ns))
        self.active = np.ones(self.B, dtype=bool)   # False once a robot's termination test has fired
        self.last_result: Optional[BatchResult] = None

    # -- per-robot set-up (same meaning as InterfaceMpc.initialization / update_*_constraints) ---------------------
    def initialization(self, i: int, init_state, goal_state, ref_path_list: Sequence[Sequence[float]], mode: str = "work"):
        self.states[i] = init_state
        self.goals[i] = goal_state
        self.last_actions[i] = 0.0
        base_speed, _ = work_mode(self.config, mode)
        self.ref_trajs[i] = global_reference_trajectory(self.config.ts, ref_path_list, self.states[i], base_speed)
        self.idx_ref[i] = 0
        self.active[i] = True

    def update_static_constraints(self, i: int, obstacle_list):
        self.stc_constraints[i] = static_obstacle_params(obstacle_list, self.config.Nstcobs, self.config.nstcobs)

    def update_dynamic_constraints(self, i: int, full_dyn_obstacle_list):
        per = self.config.ndynobs * self.config.N_hor
        block = self.dyn_constraints[i]
        for j, obstacle in enumerate(full_dyn_obstacle_list):
            block[j * per:(j + 1) * per] = [float(x) for step in obstacle for x in step]

    def share_predictions(self, groups: Optional[Sequence[Sequence[int]]] = None):
        """Fill every robot's other-robot block with the latest predictions of the robots in its group
        (layout of ``scenario_simulator.py:154-163``: robot after robot, N x (x, y, theta) each, zero padded)."""
        cfg = self.config
        groups = groups if groups is not None else [list(range(self.B))]
        for g in groups:
            for i in g:
                others = [j for j in g if j != i][:cfg.Nother]
                block = np.zeros((cfg.Nother, cfg.N_hor, cfg.ns))
                if others:
                    block[:len(others)] = self.pred_states[others]
                self.other_robot_states[i] = block.reshape(-1).tolist()

    # -- one control tick for all robots ---------------------------------------------------------------------------
    def local_refs(self) -> np.ndarray:
        """``get_local_ref_traj`` for every robot ([B, N, 3]); advances the nearest-sample indices like the reference's
        call does (``interface_mpc.py:73-80``).  Pass a (modified) copy to :meth:`step` as ``refs``."""
        cfg = self.config
        out = np.empty((self.B, cfg.N_hor, 3))
        for i in range(self.B):
            out[i], self.idx_ref[i] = local_reference_window(self.idx_ref[i], self.ref_trajs[i], self.states[i],
                                                             cfg.action_steps, cfg.N_hor)
        return out

    def assemble(self, mode: str = "work", refs: Optional[np.ndarray] = None) -> np.ndarray:
        cfg = self.config
        base_speed, tuning = work_mode(cfg, mode)
        P = np.zeros((self.B, cfg.num_params))
        if refs is None:
            refs = self.local_refs()
        for i in range(self.B):
            ref = np.asarray(refs[i], dtype=float)
            P[i] = assemble_parameters(self.states[i], ref[-1], self.last_actions[i], tuning,
                                       ref.reshape(-1), speed_references(cfg, base_speed, self.states[i], self.goals[i]),
                                       self.other_robot_states[i], self.stc_constraints[i], self.dyn_constraints[i],
                                       self.stc_weights, self.dyn_weights)
        return P

    def step(self, mode: str = "work", initial_guess: Optional[np.ndarray] = None, refs: Optional[np.ndarray] = None):
        """Solve all robots, apply the first ``action_steps`` inputs.  Returns (actions [B, nu], pred_states
        [B, N, ns], cost [B]); robots that already reached their goal keep their state (action 0).
        ``refs`` [B, N, 3]: the reference each robot tracks this tick (``get_action(current_ref_traj)``,
        ``interface_mpc.py:82-88``); default: the local window of its global reference."""
        cfg = self.config
        near = np.all(np.abs(self.states[:, :2] - self.goals[:, :2]) <= 0.05, axis=1)
        self.active &= ~(near & (np.abs(self.last_actions[:, 0]) < 0.05))     # check_termination_condition
        P = self.assemble(mode, refs)
        res = self.solver.solve(P, initial_guess)
        self.last_result = res
        u = res.solution.reshape(self.B, cfg.N_hor, cfg.nu)
        state = self.states.copy()
        for s in range(cfg.action_steps):
            state = unicycle_model(state, u[:, s], cfg.ts)
        pred = np.empty((self.B, cfg.N_hor, cfg.ns))
        rolling = state
        for k in range(cfg.N_hor):                      # rolled from the taken state, re-applying u[0] (reference quirk)
            rolling = unicycle_model(rolling, u[:, k], cfg.ts)
            pred[:, k] = rolling
        act = self.active
        self.states[act] = state[act]
        self.last_actions[act] = u[act, cfg.action_steps - 1]
        self.last_actions[~act] = 0.0
        self.pred_states[act] = pred[act]
        actions = np.where(act[:, None], u[:, 0], 0.0)
        return actions, self.pred_states.copy(), res.cost
