"""Tracker facade with the reference's API (``src/interface_mpc.py:16-88``): ``InterfaceMpc(config, use_tcp,
verbose, motion_model)``, ``initialization``, ``update_static_constraints``, ``update_dynamic_constraints``,
``update_other_robot_states``, ``get_local_ref_traj``, ``get_action`` and the read-only properties.
``TrajectoryTracker`` / ``run`` are the names BASELINE.json's north_star uses for the same thing."""
from __future__ import annotations

import itertools
from typing import Callable, List, Optional, Tuple

import numpy as np

from .geometry import static_obstacle_params
from .motion_model import unicycle_model
from .trajectory_generator import TrajectoryGenerator

DEFAULT_MOTION_MODEL = unicycle_model


class InterfaceMpc:
    def __init__(self, config, use_tcp: bool = False, verbose: bool = False, motion_model: Optional[Callable] = None,
                 solver=None, device: int = 0):
        self._traj_gen = TrajectoryGenerator(config, use_tcp, verbose=verbose, solver=solver, device=device)
        self._traj_gen.load_robot_dynamics(motion_model if motion_model is not None else DEFAULT_MOTION_MODEL)
        self._last_action = np.array([0.0, 0.0])
        self.stc_constraints = [0.0] * config.Nstcobs * config.nstcobs
        self.dyn_constraints = [0.0] * config.Ndynobs * config.ndynobs * config.N_hor
        self.other_robot_states = [0] * config.ns * config.N_hor * config.Nother

    config = property(lambda self: self._traj_gen.config)
    state = property(lambda self: self._traj_gen.state)
    last_action = property(lambda self: self._last_action)
    goal = property(lambda self: self._traj_gen.final_goal)
    ref_path = property(lambda self: self._ref_path)
    ref_traj = property(lambda self: self._traj_gen.ref_traj)

    def set_current_state(self, state: np.ndarray):
        self._traj_gen.set_current_state(state)

    def initialization(self, init_state: np.ndarray, goal_state: np.ndarray, ref_path_list: List[tuple], mode: str = "work"):
        self._ref_path = [tuple(p) for p in ref_path_list]
        self._traj_gen.load_init_state(init_state, goal_state)
        self._traj_gen.set_work_mode(mode)
        self._traj_gen.set_ref_trajectory(self._ref_path)

    def update_static_constraints(self, obstacle_list):
        block = static_obstacle_params(obstacle_list, self.config.Nstcobs, self.config.nstcobs)
        n = len(obstacle_list) * self.config.nstcobs
        self.stc_constraints[:n] = block[:n]      # slots beyond the given obstacles keep their previous content

    def update_dynamic_constraints(self, full_dyn_obstacle_list):
        per_obstacle = self.config.ndynobs * self.config.N_hor
        for i, dyn_obstacle in enumerate(full_dyn_obstacle_list):
            self.dyn_constraints[i * per_obstacle:(i + 1) * per_obstacle] = list(itertools.chain(*dyn_obstacle))

    def update_other_robot_states(self, other_robot_states):
        self.other_robot_states = other_robot_states

    def get_local_ref_traj(self, local_ref_traj: Optional[np.ndarray] = None) -> Tuple[np.ndarray, Optional[np.ndarray]]:
        original, idx = self._traj_gen.get_local_ref_traj(self._traj_gen.idx_ref, self.ref_traj, self.state,
                                                          action_steps=self.config.action_steps,
                                                          horizon=self.config.N_hor)
        self._traj_gen.idx_ref = idx
        if local_ref_traj is not None and local_ref_traj.shape[1] == 2:
            local_ref_traj = np.concatenate((local_ref_traj, original[:, [2]]), axis=1)
        return original, local_ref_traj

    def get_action(self, current_ref_traj: np.ndarray, mode: str = "work", initial_guess: Optional[np.ndarray] = None):
        if self._traj_gen.check_termination_condition(self.state, self._last_action, self.goal):
            return None
        actions, pred_states, cost = self._traj_gen.run_step(self.stc_constraints, self.dyn_constraints,
                                                             self.other_robot_states, current_ref_traj, mode,
                                                             initial_guess)
        self._last_action = actions[0]
        return actions[0], pred_states, cost

    run = get_action


TrajectoryTracker = InterfaceMpc
