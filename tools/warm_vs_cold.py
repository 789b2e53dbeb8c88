"""Cold start (u0 = 0, the reference's call sites) vs warm start (previous solution shifted by one step) on the benchmark
scenes advanced by one control tick.   usage: python tools/warm_vs_cold.py [B]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from trajtrack_mpcndqn_rlboost_amd import BatchSolver, MpcConfig, scenes
from trajtrack_mpcndqn_rlboost_amd.motion_model import unicycle_model

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
cfg = MpcConfig()
bs = BatchSolver(cfg)
sc = scenes.make_batch(cfg, B, n_dyn=8, seed=1234)
p = sc["p"].copy()
cold = bs.solve(p)
t_cold = bs.last_timing()["solve_ms"]
# advance every robot by its first input (state, last input), keep the rest of the scene: the next tick's problem
u = cold.solution.reshape(B, cfg.N_hor, 2)
p[:, 0:3] = unicycle_model(p[:, 0:3], u[:, 0], cfg.ts)
p[:, 6:8] = u[:, 0]
cold2 = bs.solve(p)
t_cold2 = bs.last_timing()["solve_ms"]
warm = bs.solve(p, scenes.shifted_warm_start(cold.solution))
t_warm = bs.last_timing()["solve_ms"]
for name, r, t in (("cold, tick 0", cold, t_cold), ("cold, tick 1", cold2, t_cold2), ("warm, tick 1", warm, t_warm)):
    print(f"{name}: {B / t * 1e3:8.0f} solves/s  kernel {t:7.1f} ms  mean inner it {r.num_inner_iterations.mean():6.0f}  "
          f"converged {np.mean(r.status == 0):.3f}  median cost {np.median(r.cost):.1f}")
