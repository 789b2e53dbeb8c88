"""The reference's solver-plugin contract on top of libmpcgpu.so.

The reference imports a generated PyO3 module named ``config.optimizer_name`` and calls
``module.solver().run(p, initial_guess)`` (``src/mpc_traj_tracker/trajectory_generator.py:63-71,318``; stub
signature with ``initial_lagrange_multipliers`` / ``initial_penalty`` at ``:25-27``), then reads
``.solution, .cost, .exit_status, .solve_time_ms`` (``:320-323``).  ``Solver`` honours that contract for a
single problem (B = 1 through the same batched kernel) and adds ``run_batch``.

Error behaviour follows the OpEn binding: wrong input lengths print the binding's diagnostic and return
``None``; a failing library call raises ``RuntimeError`` (what the reference's callers catch,
``trajectory_generator.py:277-282``).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np

from .config import MpcConfig
from .solver import BatchSolver, BatchResult, STATUS_NAMES


class SolverStatus:
    """Attribute-compatible with OpEn's ``OptimizerSolution``."""

    def __init__(self, res: BatchResult, i: int):
        self.exit_status: str = STATUS_NAMES[int(res.status[i])]
        self.num_outer_iterations: int = int(res.num_outer_iterations[i])
        self.num_inner_iterations: int = int(res.num_inner_iterations[i])
        self.last_problem_norm_fpr: float = float(res.last_problem_norm_fpr[i])
        self.f2_norm: float = float(res.f2_norm[i])
        self.solve_time_ms: float = float(res.solve_time_ms[i])
        self.solution: List[float] = res.solution[i].tolist()
        self.lagrange_multipliers: List[float] = res.lagrange_multipliers[i].tolist()
        self.cost: float = float(res.cost[i])

    def __repr__(self):
        return (f"SolverStatus(exit_status={self.exit_status!r}, cost={self.cost:.6g}, "
                f"inner={self.num_inner_iterations}, outer={self.num_outer_iterations})")


class Solver:
    def __init__(self, config: Optional[MpcConfig] = None, device: int = 0):
        self._batch = BatchSolver(config, device=device)

    @property
    def batch_solver(self) -> BatchSolver:
        return self._batch

    def run(self, p: Sequence[float], initial_guess=None, initial_lagrange_multipliers=None,
            initial_penalty: Optional[float] = None) -> Optional[SolverStatus]:
        bs = self._batch
        p = np.asarray(p, dtype=np.float64).reshape(-1)
        if initial_guess is not None and np.size(initial_guess) != bs.n:
            print("1600 -> Initial guess has incompatible dimensions")
            return None
        if initial_lagrange_multipliers is not None and np.size(initial_lagrange_multipliers) != bs.n:
            print("1700 -> wrong dimension of Langrange multipliers")
            return None
        if p.size != bs.np:
            print("3003 -> wrong number of parameters")
            return None
        res = bs.solve(
            p[None],
            None if initial_guess is None else np.asarray(initial_guess, dtype=np.float64).reshape(1, -1),
            None if initial_lagrange_multipliers is None else
            np.asarray(initial_lagrange_multipliers, dtype=np.float64).reshape(1, -1),
            None if initial_penalty is None else np.array([float(initial_penalty)]))
        if int(res.status[0]) == 3:      # OpEn: Err(NotFiniteComputation) -> the binding prints and returns None
            print("2000 -> Problem solution failed (solver error)")
            return None
        return SolverStatus(res, 0)

    def run_batch(self, p, initial_guess=None, initial_lagrange_multipliers=None, initial_penalty=None) -> BatchResult:
        return self._batch.solve(p, initial_guess, initial_lagrange_multipliers, initial_penalty)


_DEFAULT_CONFIG: Optional[MpcConfig] = None


def set_default_config(config: MpcConfig) -> None:
    global _DEFAULT_CONFIG
    _DEFAULT_CONFIG = config


def solver(config: Optional[MpcConfig] = None, device: int = 0) -> Solver:
    """``built_solver.solver()`` of the reference (trajectory_generator.py:71)."""
    return Solver(config if config is not None else _DEFAULT_CONFIG, device=device)
