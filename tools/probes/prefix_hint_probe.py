#!/usr/bin/env python3
"""How well do the first k inner problems of a solve predict its total length?  Benchmark family, B problems: psi-evaluation counts
with the outer cap at 1, 2, 3 (run-time caps: the same iteration, cut short) and of the full solve -> gpurun_out/prefix_hint_<B>.npz
(analysed offline by tools/probes/prefix_hint_analysis.py with the fluid model of the launch)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from trajtrack_mpcndqn_rlboost_amd import MpcConfig, BatchSolver, scenes
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
fam = sys.argv[2] if len(sys.argv) > 2 else "bench"
base = MpcConfig()
sc = scenes.make_batch(base, B, n_dyn=8, seed=1236) if fam == "bench" else scenes.make_family(base, B, fam, n_dyn=8, seed=1236)
rec = {}
for mo in (1, 2, 3, 10):
    bs = BatchSolver(MpcConfig(solver_max_outer_iterations=mo), order="as_given", tail_promotion=0)
    r = bs.solve(sc["p"])
    ev = bs.last_eval_counts(B)
    rec[f"evals_{mo}"] = ev[0].copy(); rec[f"inner_{mo}"] = r.num_inner_iterations.copy(); rec[f"status_{mo}"] = r.status.copy()
    rec[f"outer_{mo}"] = r.num_outer_iterations.copy(); rec[f"ms_{mo}"] = bs.last_timing()["solve_ms"]
    print(mo, "kernel ms", rec[f"ms_{mo}"], "mean evals", ev[0].mean(), flush=True)
    bs.close()
np.savez_compressed(os.path.join(ROOT, "gpurun_out", f"prefix_hint_{fam}_{B}.npz"), **rec)
