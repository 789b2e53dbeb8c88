"""Tracker harness around the solver: reference trajectory, parameter assembly, post-solve rollouts.

Mirror of the reference's ``TrajectoryGenerator`` (``src/mpc_traj_tracker/trajectory_generator.py:30-339``): same
method names, argument meaning and return values, so callers written against the reference run unchanged; the
solver behind it is the MI355X library (``plugin.Solver``) instead of the generated OpEn module.  The pure
functions (`global_reference_trajectory`, `local_reference_window`, `speed_references`, `assemble_parameters`,
`rollout_after_solve`) are shared with the batched tracker.
"""
from __future__ import annotations

import math
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

from .config import MpcConfig
from .motion_model import unicycle_model

N_TUNING = 10


# ---------------------------------------------------------------------------------------------------------
# pure functions
# ---------------------------------------------------------------------------------------------------------
def work_mode(config: MpcConfig, mode: str) -> Tuple[float, List[float]]:
    """(base_speed, tuning_params) of a work mode -- trajectory_generator.py:115-138."""
    if mode == "aligning":
        weights = [0.0] * N_TUNING
        weights[2] = 100
        return config.lin_vel_max * config.medium_speed, weights
    weights = [config.qpos, config.qvel, config.qtheta, config.lin_vel_penalty, config.ang_vel_penalty,
               config.qpN, config.qthetaN, config.qrpd, config.lin_acc_penalty, config.ang_acc_penalty]
    fraction = {"safe": config.low_speed, "work": config.high_speed, "super": config.full_speed}
    if mode not in fraction:
        raise ModuleNotFoundError(f"There is no mode called {mode}.")
    return config.lin_vel_max * fraction[mode], weights


def global_reference_trajectory(ts: float, ref_path: Sequence[Sequence[float]], state: Sequence[float],
                                speed: float) -> np.ndarray:
    """Sample the reference polyline every ``speed * ts`` metres starting at ``state``; rows (x, y, heading).

    Reproduces trajectory_generator.py:165-204 step for step, including its corner behaviour: when a node is
    reached inside a sampling period the walker snaps to the node and then takes a FULL step along the next leg
    (the remaining time of the period is not carried over), and a node closer than 1e-9 is skipped without
    emitting a sample."""
    x, y = float(state[0]), float(state[1])
    nodes = [(float(n[0]), float(n[1])) for n in ref_path]
    target = 0
    rows: List[Tuple[float, float, float]] = []
    step = speed * ts
    travelling = True
    x_dir = y_dir = 0.0
    while travelling:
        emit = True
        while True:
            tx, ty = nodes[target]
            dist = math.hypot(tx - x, ty - y)
            if dist < 1e-9:                       # already on the node: aim at the next one, no sample
                target += 1
                emit = False
                break
            x_dir, y_dir = (tx - x) / dist, (ty - y) / dist
            eta = dist / speed
            if eta > ts:                          # a full sampling period fits on this leg
                x, y = x + x_dir * speed * ts, y + y_dir * speed * ts
                break
            x, y = x + x_dir * speed * eta, y + y_dir * speed * eta   # snap to the node
            target += 1
            if target > len(nodes) - 1:
                travelling = False
                break
        if emit:
            rows.append((x, y, math.atan2(y_dir, x_dir)))
    return np.array(rows, dtype=float).reshape(-1, 3)


def local_reference_window(idx_ref: int, ref_traj_global: np.ndarray, state: Sequence[float], action_steps: int = 1,
                           horizon: int = 20) -> Tuple[np.ndarray, int]:
    """Nearest reference sample in the window [idx-1, idx+5) action steps, then ``horizon`` rows from there, the
    tail padded with the last sample -- trajectory_generator.py:206-232."""
    n = len(ref_traj_global)
    lb = max(0, idx_ref - 1 * action_steps)
    ub = min(n, idx_ref + 5 * action_steps)
    d = [math.hypot(state[0] - r[0], state[1] - r[1]) for r in ref_traj_global[lb:ub]]
    idx_next = d.index(min(d)) + lb
    if idx_next + horizon >= n:
        pad = horizon - (n - idx_next)
        window = np.concatenate([ref_traj_global[idx_next:], np.repeat(ref_traj_global[-1:], pad, axis=0)], axis=0)
    else:
        window = ref_traj_global[idx_next:idx_next + horizon]
    return np.array(window, dtype=float), idx_next


def speed_references(config: MpcConfig, base_speed: float, state: Sequence[float], final_goal: Sequence[float]) -> List[float]:
    """Constant speed reference, scaled down near the goal (floor: the *fraction* ``low_speed`` used as m/s) --
    trajectory_generator.py:257-264."""
    N, ts = config.N_hor, config.ts
    dist_to_goal = math.hypot(state[0] - final_goal[0], state[1] - final_goal[1])
    if dist_to_goal >= base_speed * N * ts:
        return [base_speed] * N
    return [max(dist_to_goal / N / ts, config.low_speed)] * N


def assemble_parameters(state, finish_state, last_u, tuning_params, current_refs, speed_refs, other_robot_states,
                        stc_constraints, dyn_constraints, stc_weights, dyn_weights) -> List[float]:
    """Parameter vector in the order of mpc_generator.py:179-188 -- trajectory_generator.py:272-275."""
    return list(state) + list(finish_state) + list(last_u) + list(tuning_params) + list(current_refs) + \
        list(speed_refs) + list(other_robot_states) + list(stc_constraints) + list(dyn_constraints) + \
        list(stc_weights) + list(dyn_weights)


def rollout_after_solve(motion_model: Callable, state: np.ndarray, u: Sequence[float], nu: int, ts: float,
                        take_steps: int):
    """taken states, predicted states (rolled from the TAKEN state, re-applying u[0] -- the reference's quirk) and
    the applied actions -- trajectory_generator.py:325-338."""
    u = np.asarray(u, dtype=float)
    taken = [motion_model(state, u[i * nu:(i + 1) * nu], ts) for i in range(take_steps)]
    pred = [taken[-1]]
    for i in range(len(u) // nu):
        pred.append(motion_model(pred[-1], u[i * nu:(i + 1) * nu], ts))
    actions = [np.array(a) for a in u[:nu * take_steps].reshape(take_steps, nu).tolist()]
    return taken, pred[1:], actions


# ---------------------------------------------------------------------------------------------------------
# single-robot harness with the reference's API
# ---------------------------------------------------------------------------------------------------------
class TrajectoryGenerator:
    def __init__(self, config: MpcConfig, use_tcp: bool = False, verbose: bool = False, solver=None, device: int = 0):
        if use_tcp:
            raise NotImplementedError("the TCP transport of the OpEn server is not part of this build")
        self._prtname = "[Traj]"
        self.vb = verbose
        self.config = config
        self.ts, self.ns, self.nu, self.N_hor = config.ts, config.ns, config.nu, config.N_hor
        self.use_tcp = False
        self.set_work_mode(mode="safe")
        self.set_obstacle_weights(stc_weights=1e3, dyn_weights=1e3)
        if solver is None:
            from .plugin import Solver
            solver = Solver(config, device=device)
        self.solver = solver
        self.motion_model: Callable = unicycle_model

    # -- configuration ------------------------------------------------------------------------------------
    def load_robot_dynamics(self, motion_model: Callable) -> None:
        self.motion_model = motion_model

    def load_init_state(self, current_state: np.ndarray, goal_state: np.ndarray):
        if not isinstance(current_state, np.ndarray) or not isinstance(goal_state, np.ndarray):
            raise TypeError(f"State and action should be numpy.ndarry, got {type(current_state)}/{type(goal_state)}.")
        self.state = current_state
        self.final_goal = goal_state
        self.past_states, self.past_actions = [], []
        self.cost_timelist, self.solver_time_timelist = [], []
        self.idx_ref = 0

    def set_obstacle_weights(self, stc_weights, dyn_weights):
        def expand(w):
            if isinstance(w, list):
                return w
            if isinstance(w, (float, int)):
                return [w] * self.N_hor
            raise TypeError(f"Unsupported datatype for obstacle weights, got {type(w)}.")
        self.stc_weights, self.dyn_weights = expand(stc_weights), expand(dyn_weights)

    def set_work_mode(self, mode: str = "safe"):
        self.base_speed, self.tuning_params = work_mode(self.config, mode)

    def set_current_state(self, current_state: np.ndarray):
        if not isinstance(current_state, np.ndarray):
            raise TypeError(f"State should be numpy.ndarry, got {type(current_state)}.")
        self.state = current_state

    def set_ref_trajectory(self, ref_path):
        self.idx_ref = 0
        self.ref_traj = self.get_global_ref_traj(self.ts, ref_path, self.state, self.base_speed)

    def check_termination_condition(self, state, action, final_goal) -> bool:
        done = bool(np.allclose(state[:2], final_goal[:2], atol=0.05, rtol=0) and abs(action[0]) < 0.05)
        if done:
            print(f"{self._prtname} MPC solution found.")
        return done

    # -- reference trajectory -------------------------------------------------------------------------------
    @staticmethod
    def get_global_ref_traj(ts, ref_path, state, speed) -> np.ndarray:
        return global_reference_trajectory(ts, ref_path, state, speed)

    @staticmethod
    def get_local_ref_traj(idx_ref, ref_traj_global, state, action_steps=1, horizon=20):
        return local_reference_window(idx_ref, np.asarray(ref_traj_global, dtype=float), state, action_steps, horizon)

    # -- one control step ---------------------------------------------------------------------------------------
    def run_step(self, stc_constraints: list, dyn_constraints: list, other_robot_states: list,
                 current_ref_traj: np.ndarray, mode: str = "safe", initial_guess: Optional[np.ndarray] = None):
        self.set_work_mode(mode)
        finish_state = current_ref_traj[-1, :]
        speed_refs = speed_references(self.config, self.base_speed, self.state, self.final_goal)
        last_u = self.past_actions[-1] if len(self.past_actions) else np.zeros(self.nu)
        params = assemble_parameters(self.state, finish_state, last_u, self.tuning_params,
                                     current_ref_traj.reshape(-1).tolist(), speed_refs, other_robot_states,
                                     stc_constraints, dyn_constraints, self.stc_weights, self.dyn_weights)
        try:
            taken, pred, actions, cost, solver_time, exit_status = self.run_solver(
                params, self.state, self.config.action_steps, initial_guess)
        except RuntimeError as err:
            raise RuntimeError(f"Fatal: Cannot run solver. {err}.")
        self.past_states.append(self.state)
        self.past_states += taken[:-1]
        self.past_actions += actions
        self.state = taken[-1]
        self.cost_timelist.append(cost)
        self.solver_time_timelist.append(solver_time)
        if exit_status in self.config.bad_exit_codes and self.vb:
            print(f"{self._prtname} Bad converge status: {exit_status}")
        return actions, pred, cost

    def run_solver(self, parameters: list, state: np.ndarray, take_steps: int = 1,
                   initial_guess: Optional[np.ndarray] = None):
        solution = self.solver.run(parameters, initial_guess)
        if solution is None:
            raise RuntimeError("the solver rejected its inputs (see the diagnostic printed above)")
        taken, pred, actions = rollout_after_solve(self.motion_model, state, solution.solution, self.nu, self.ts,
                                                   take_steps)
        return taken, pred, actions, solution.cost, solution.solve_time_ms, solution.exit_status
