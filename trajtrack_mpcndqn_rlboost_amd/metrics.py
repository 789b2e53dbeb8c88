"""Evaluation metrics of a run, as the reference's ``Metrics`` (``src/main_pre.py:55-144``) computes them for
``src/main_evaluation.py:262-320``: computation time (mean / max / median), deviation from the reference trajectory
(mean / max of the distance to the nearest reference sample), action smoothness (mean absolute second difference of
speed and angular speed), minimal clearance to the (un-inflated) static obstacles, finish time in steps (successful runs
only) and success rate.  Same class name, method names, arguments and result dictionary; array arithmetic instead of
shapely, plus :meth:`add_batch` for the batched loop (``hybrid.BatchedHybrid.run(record=True)``)."""
from __future__ import annotations

import statistics
from typing import Dict, List, Sequence

import numpy as np

from .hybrid import polygon_distance


class Metrics:
    def __init__(self, mode: str) -> None:
        if mode not in ("dqn", "mpc", "hyb"):
            raise ValueError(f"Mode {mode} not recognized (should be 'dqn', 'mpc', or 'hyb').")
        self.mode = mode
        self.trial_list: List[Dict] = []
        self.success_rate = 0

    # ---- per-trial quantities ------------------------------------------------------------------------------------------
    @staticmethod
    def _get_computation_time(times: Sequence[float]):
        return [statistics.mean(times), max(times), statistics.median(times)]

    @staticmethod
    def _get_deviation_distance(ref_traj, actual_traj):
        ref = np.asarray(ref_traj, dtype=float)[:, :2]
        act = np.asarray(actual_traj, dtype=float)[:, :2]
        d = np.hypot(act[:, None, 0] - ref[None, :, 0], act[:, None, 1] - ref[None, :, 1]).min(axis=1)
        return [statistics.mean(d.tolist()), float(d.max())]

    @staticmethod
    def _get_smoothness(action_list):
        a = np.asarray(action_list, dtype=float)
        return [statistics.mean(np.abs(np.diff(a[:, 0], n=2)).tolist()), statistics.mean(np.abs(np.diff(a[:, 1], n=2)).tolist())]

    @staticmethod
    def _get_minimal_obstacle_distance(trajectory, obstacles):
        return min(min(polygon_distance(ob, pos) for ob in obstacles) for pos in trajectory)

    @staticmethod
    def _get_finish_time_steps(times, succeed: bool):
        return len(times) if succeed else -1

    def _get_success_rate(self):
        self.success_rate = sum(t["success"] for t in self.trial_list) / len(self.trial_list)

    # ---- API of the reference --------------------------------------------------------------------------------------------
    def add_trial_result(self, computation_time_list, succeed: bool, action_list, ref_trajectory, actual_trajectory,
                         obstacle_list):
        m = {"computation_time": self._get_computation_time(computation_time_list),
             "deviation_distance": self._get_deviation_distance(ref_trajectory, actual_trajectory),
             "smoothness": self._get_smoothness(action_list),
             "clearance": self._get_minimal_obstacle_distance(actual_trajectory, obstacle_list),
             "finish_time": self._get_finish_time_steps(computation_time_list, succeed)}
        m["success"] = m["finish_time"] > 0
        self.trial_list.append(m)
        self._get_success_rate()

    def get_average(self, round_digit: int = 4) -> dict:
        mean = lambda xs: round(statistics.mean(xs), round_digit)   # noqa: E731
        finish = [t["finish_time"] for t in self.trial_list if t["success"]] or [-1]
        self.metric_average = {
            "computation_time": [mean([t["computation_time"][k] for t in self.trial_list]) for k in range(3)],
            "deviation_distance": [mean([t["deviation_distance"][k] for t in self.trial_list]) for k in range(2)],
            "smoothness": [mean([t["smoothness"][k] for t in self.trial_list]) for k in range(2)],
            "clearance": mean([t["clearance"] for t in self.trial_list]),
            "finish_time": mean(finish),
            "success_rate": self.success_rate,
        }
        return self.metric_average

    # ---- batched loop ------------------------------------------------------------------------------------------------------
    def add_batch(self, record: Dict, obstacle_lists: Sequence[Sequence]) -> None:
        """One trial per robot of a ``BatchedHybrid.run(record=True)`` result: ``record`` holds, per robot, the tick times
        [ms], the success flag, the (v, w) history, the global reference trajectory and the traversed positions."""
        for b in range(len(record["success"])):
            self.add_trial_result(record["tick_ms"][b], bool(record["success"][b]), record["actions"][b],
                                  record["ref_traj"][b], record["positions"][b], obstacle_lists[b])
