#!/usr/bin/env python3
"""One batched solve on a chosen scene; prints the evaluation counts so that `rocprofv3 --pmc SQ_INSTS_VALU` can be
turned into VALU instructions per psi evaluation.  usage: valu_per_eval.py B n_dyn with_static(0/1) [N]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from trajtrack_mpcndqn_rlboost_amd import MpcConfig, BatchSolver, scenes
B, n_dyn, st = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
N = int(sys.argv[4]) if len(sys.argv) > 4 else 20
cfg = MpcConfig(N_hor=N)
bs = BatchSolver(cfg)
sc = scenes.make_batch(cfg, B, n_dyn=n_dyn, seed=1236, with_box=bool(st), with_walls=bool(st))
res = bs.solve(sc["p"])
n_psi, n_grad = bs.last_eval_counts(B)
print(f"EVALS total_psi {int(n_psi.sum())} total_grad {int(n_grad.sum())} inner {int(res.num_inner_iterations.sum())} solve_ms {bs.last_timing()['solve_ms']:.1f}")
