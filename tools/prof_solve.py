#!/usr/bin/env python3
"""Run a few batched solves (for rocprofv3). usage: prof_solve.py [B] [reps] [n_dyn] [N]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from trajtrack_mpcndqn_rlboost_amd import MpcConfig, BatchSolver, scenes
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
n_dyn = int(sys.argv[3]) if len(sys.argv) > 3 else 8
N = int(sys.argv[4]) if len(sys.argv) > 4 else 20
cfg = MpcConfig(N_hor=N)
bs = BatchSolver(cfg)
sc = scenes.make_batch(cfg, B, n_dyn=n_dyn, seed=1236)
for _ in range(reps):
    res = bs.solve(sc["p"])
    print("solve_ms", bs.last_timing(), "inner", res.num_inner_iterations.mean(), "shape", bs.last_shape())
