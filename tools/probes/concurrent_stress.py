#!/usr/bin/env python3
"""Stress run of the tail promotion with the continuation beside the draining launch (MPCGPU_OPT_TAIL_CONCURRENT, the default): random
horizons, batch sizes, scene families, dispatch orders and starts, several calls per handle (the list, the counters and the side
stream are reused from call to call) -- every output of every call must be BITWISE what a handle without promotion writes.
usage: python tools/probes/concurrent_stress.py [trials = 24] [seed = 0]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from trajtrack_mpcndqn_rlboost_amd import MpcConfig, BatchSolver, scenes

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
FIELDS = ("solution", "cost", "status", "num_inner_iterations", "num_outer_iterations", "last_problem_norm_fpr", "f2_norm", "lagrange_multipliers")
bad, promoted_total, calls = 0, 0, 0
t0 = time.time()
for trial in range(trials):
    N = int(rng.choice([20, 20, 40]))
    cfg = MpcConfig(N_hor=N, solver_max_inner_iterations=int(rng.choice([500, 200, 60])), solver_max_outer_iterations=int(rng.choice([10, 6])))
    order = str(rng.choice(["as_given", "longest_first"]))
    ref = BatchSolver(cfg, latency_batch=0, order=order, tail_promotion=0)
    tst = BatchSolver(cfg, latency_batch=0, order=order)          # defaults: promotion on, continuation concurrent where it applies
    for call in range(3):                                          # the same two handles, another batch each time
        B = int(rng.integers(1100, 9000 if N == 20 else 5000))
        fam = str(rng.choice(["benchmark", "passing", "avoidance"]))
        sc = scenes.make_family(cfg, B, fam, n_dyn=int(rng.integers(4, 9)), seed=int(rng.integers(1 << 30)))
        kw = {}
        if rng.random() < 0.5:
            kw["initial_guess"] = rng.normal(0.0, 0.3, (B, 2 * N))
        a, b = ref.solve(sc["p"], **kw), tst.solve(sc["p"], **kw)
        cap, moved = tst.last_tail_promotion()
        same = all(np.array_equal(getattr(a, f), getattr(b, f), equal_nan=True) for f in FIELDS)
        ea, eb = ref.last_eval_counts(B), tst.last_eval_counts(B)
        same = same and np.array_equal(ea[0], eb[0]) and np.array_equal(ea[1], eb[1])
        calls += 1; promoted_total += moved; bad += 0 if same else 1
        print(f"trial {trial:2d}.{call} N {N} B {B:5d} {fam:9s} {order:13s} start {'given' if kw else 'cold '}: capacity {cap:4d}, {moved:4d} promoted, "
              f"converged {int(np.sum(a.status == 0)):5d}, {'same bits' if same else 'DIFFERENT'}", flush=True)
    ref.close(); tst.close()
print(f"{calls} calls, {promoted_total} problems promoted, {bad} calls with different bits ({time.time() - t0:.0f} s)")
sys.exit(1 if bad else 0)
