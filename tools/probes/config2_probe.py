import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from trajtrack_mpcndqn_rlboost_amd import MpcConfig, BatchSolver, scenes
for stall in ("either", "both"):
    cfg = MpcConfig(N_hor=20, solver_penalty_stall=stall)
    sc = scenes.make_batch(cfg, 1024, n_dyn=4, seed=1236)
    for lat in (None, 0):
        bs = BatchSolver(cfg, latency_batch=lat)
        for _ in range(2):
            res = bs.solve(sc["p"])
        t = bs.last_timing()
        n_psi, n_grad = bs.last_eval_counts(1024)
        print(f"{stall:6s} latency_batch={lat}: solve {t['solve_ms']:.1f} ms; inner mean {res.num_inner_iterations.mean():.0f} max {res.num_inner_iterations.max()}; psi evals mean {n_psi.mean():.0f} "
              f"max {n_psi.max()} p99 {np.quantile(n_psi, 0.99):.0f}; evals per step mean {n_psi.sum() / max(res.num_inner_iterations.sum(), 1):.2f}, of the slowest problem "
              f"{n_psi[np.argmax(res.solve_time_ms)] / max(res.num_inner_iterations[np.argmax(res.solve_time_ms)], 1):.2f} ({res.solve_time_ms.max():.1f} ms); latency kernel {bs.last_shape()['latency_kernel']}", flush=True)
        bs.close()
