#!/usr/bin/env python3
"""Single-call latency of the reference's call pattern: ONE problem per `solver.run(p)` (src/interface_mpc.py:82-88).
Wall time around plugin.Solver.run (host buffers in, host results out) and the kernel's own time, latency kernel vs
throughput kernel, on an easy scene (converges in a few dozen iterations) and on a scene that runs into the iteration caps.
usage: latency.py [reps]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from trajtrack_mpcndqn_rlboost_amd import MpcConfig, BatchSolver, Solver, scenes

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cfg = MpcConfig()
easy = scenes.make_batch(cfg, 8, n_dyn=0, with_box=False, seed=8, v_init_range=(1.0, 1.2))["p"]
hard = scenes.make_batch(cfg, 8, n_dyn=8, seed=1236)["p"]
print(f"mpc_default.yaml, N_hor = 20; {reps} calls each; times in ms")
for name, lat in (("latency kernel", None), ("throughput kernel", 0)):
    bs = BatchSolver(cfg, latency_batch=lat)
    plug = Solver(cfg, batch=bs) if "batch" in Solver.__init__.__code__.co_varnames else None
    for scene, P in (("easy", easy), ("cap-hitting", hard)):
        for B in (1, 8):
            wall, kern, its = [], [], []
            for r in range(reps):
                p = P[:B] if B > 1 else P[r % 8]
                t0 = time.perf_counter()
                res = bs.solve(p)
                wall.append((time.perf_counter() - t0) * 1e3)
                kern.append(bs.last_timing()["solve_ms"] + bs.last_timing()["prep_ms"])
                its.append(res.num_inner_iterations.max())
            wall, kern = np.array(wall[2:]), np.array(kern[2:])
            print(f"  {name:18s} {scene:12s} B={B}: call {np.median(wall):8.3f} (min {wall.min():8.3f})  kernels {np.median(kern):8.3f}  "
                  f"max inner iterations {int(np.max(its))}  status {np.bincount(res.status, minlength=3).tolist()}")
    bs.close()
