"""Single-robot tracker facade.

API-compatible with the reference's ``InterfaceMpc`` (``src/interface_mpc.py:16-88``): same constructor arguments,
same method names, same return values, so ``main.py`` / ``scenario_simulator.py``-style callers work unchanged.
Internally the three padded parameter blocks (static obstacles, dynamic obstacles, other robots) are float64 numpy
buffers that are handed to the harness as views; nothing is re-allocated per control step.
``TrajectoryTracker`` and ``run`` are the names BASELINE.json's north_star uses for this class / ``get_action``.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

from .geometry import static_obstacle_params
from .motion_model import unicycle_model
from .trajectory_generator import TrajectoryGenerator


class InterfaceMpc:
    """Facade over :class:`TrajectoryGenerator` that owns the padded constraint buffers of one robot."""

    def __init__(self, config, use_tcp: bool = False, verbose: bool = False, motion_model: Optional[Callable] = None,
                 solver=None, device: int = 0):
        harness = TrajectoryGenerator(config, use_tcp, verbose=verbose, solver=solver, device=device)
        harness.load_robot_dynamics(unicycle_model if motion_model is None else motion_model)
        self._traj_gen = harness
        self._last_action = np.zeros(config.nu)
        self._ref_path: List[Tuple[float, ...]] = []
        n_hor = config.N_hor
        self._stc = np.zeros(config.Nstcobs * config.nstcobs)
        self._dyn = np.zeros(config.Ndynobs * config.ndynobs * n_hor)
        self._others = np.zeros(config.ns * n_hor * config.Nother)

    # ---- read-only views the reference exposes as properties ---------------------------------------------------------
    @property
    def config(self):
        return self._traj_gen.config

    @property
    def state(self) -> np.ndarray:
        return self._traj_gen.state

    @property
    def last_action(self) -> np.ndarray:
        return self._last_action

    @property
    def goal(self) -> np.ndarray:
        return self._traj_gen.final_goal

    @property
    def ref_path(self):
        return self._ref_path

    @property
    def ref_traj(self) -> np.ndarray:
        return self._traj_gen.ref_traj

    # the reference keeps these three blocks as public list attributes; expose them the same way
    @property
    def stc_constraints(self) -> List[float]:
        return self._stc.tolist()

    @property
    def dyn_constraints(self) -> List[float]:
        return self._dyn.tolist()

    @property
    def other_robot_states(self) -> List[float]:
        return self._others.tolist()

    # ---- set-up ------------------------------------------------------------------------------------------------------
    def set_current_state(self, state: np.ndarray) -> None:
        self._traj_gen.set_current_state(state)

    def initialization(self, init_state: np.ndarray, goal_state: np.ndarray, ref_path_list: Sequence[Sequence[float]],
                       mode: str = "work") -> None:
        """Start an episode: state, goal, reference polyline (list of (x, y[, theta]) nodes) and work mode."""
        self._ref_path = [tuple(float(c) for c in node) for node in ref_path_list]
        self._last_action = np.zeros(self.config.nu)
        self._traj_gen.load_init_state(init_state, goal_state)
        self._traj_gen.set_work_mode(mode)
        self._traj_gen.set_ref_trajectory(self._ref_path)

    def update_static_constraints(self, obstacle_list) -> None:
        """Polygons (vertex lists) -> half-space rows in the first ``len(obstacle_list)`` slots; later slots keep
        whatever they held (the reference overwrites slot by slot, ``interface_mpc.py:60-63``)."""
        cfg = self.config
        rows = static_obstacle_params(obstacle_list, cfg.Nstcobs, cfg.nstcobs)
        used = len(obstacle_list) * cfg.nstcobs
        self._stc[:used] = rows[:used]

    def update_dynamic_constraints(self, full_dyn_obstacle_list) -> None:
        """``full_dyn_obstacle_list[i][k]`` = (x, y, rx, ry, angle, alpha) of obstacle i at prediction step k."""
        cfg = self.config
        width = cfg.ndynobs * cfg.N_hor
        for slot, prediction in enumerate(full_dyn_obstacle_list):
            flat = np.asarray(prediction, dtype=float).reshape(-1)
            self._dyn[slot * width:slot * width + flat.size] = flat

    def update_other_robot_states(self, other_robot_states) -> None:
        block = np.asarray(other_robot_states, dtype=float).reshape(-1)
        if block.size != self._others.size:
            raise ValueError(f"other_robot_states must have {self._others.size} entries, got {block.size}")
        self._others = block.copy()

    # ---- per control step ---------------------------------------------------------------------------------------------
    def get_local_ref_traj(self, local_ref_traj: Optional[np.ndarray] = None):
        """(window of the global reference starting at the nearest sample, `local_ref_traj` with the window's heading
        column appended when it only has x and y) -- ``interface_mpc.py:73-80``."""
        cfg, harness = self.config, self._traj_gen
        window, nearest = harness.get_local_ref_traj(harness.idx_ref, self.ref_traj, self.state,
                                                     action_steps=cfg.action_steps, horizon=cfg.N_hor)
        harness.idx_ref = nearest
        proposal = local_ref_traj
        if proposal is not None and proposal.shape[1] == 2:
            proposal = np.hstack([proposal, window[:, 2:3]])
        return window, proposal

    def get_action(self, current_ref_traj: np.ndarray, mode: str = "work", initial_guess: Optional[np.ndarray] = None):
        """One MPC step.  ``None`` once the goal is reached, else (first action, predicted states, cost)."""
        harness = self._traj_gen
        if harness.check_termination_condition(self.state, self._last_action, self.goal):
            return None
        actions, predicted, cost = harness.run_step(self._stc.tolist(), self._dyn.tolist(), self._others.tolist(),
                                                    current_ref_traj, mode, initial_guess)
        self._last_action = actions[0]
        return actions[0], predicted, cost

    run = get_action


TrajectoryTracker = InterfaceMpc
