"""TEST INFRASTRUCTURE: a stand-in for ``BatchSolver`` that answers through the CPU oracle.

Used where no GPU is present (the build container) to drive code that sits ABOVE the solver boundary -- the plugin
module ``mpc_solver/<optimizer_name>``, the tracker harness -- end to end.  Tests only; the product never imports it."""
import numpy as np

import oracle
from trajtrack_mpcndqn_rlboost_amd.solver import BatchResult


class OracleBatchSolver:
    def __init__(self, config=None, device: int = 0, library=None):
        from trajtrack_mpcndqn_rlboost_amd import MpcConfig
        self.config = config if config is not None else MpcConfig()
        self._ocfg = oracle.OracleConfig.from_dict(self.config.solver_dict())
        self.N = int(self.config.N_hor)
        self.n = 2 * self.N
        self.np = int(self.config.num_params)
        self.calls = 0

    def solve(self, p, initial_guess=None, initial_lagrange_multipliers=None, initial_penalty=None) -> BatchResult:
        self.calls += 1
        p = np.ascontiguousarray(p, dtype=np.float64)
        if p.ndim == 1:
            p = p[None]
        u, y, r, _ = oracle.solve_batch(self._ocfg, p, initial_guess, initial_lagrange_multipliers, initial_penalty,
                                        nthreads=1)
        return BatchResult(u, r["cost"].copy(), r["status"].copy(), r["inner_iters"].copy(), r["outer_iters"].copy(),
                           r["fpr"].copy(), r["f2_norm"].copy(), y, r["solve_time_ms"].copy())

    def close(self):
        pass
