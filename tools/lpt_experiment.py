import sys, numpy as np
sys.path.insert(0, '/root/repo')
from trajtrack_mpcndqn_rlboost_amd import MpcConfig, BatchSolver, scenes
cfg = MpcConfig(N_hor=20)
for B in (8192, 32768):
    sc = scenes.make_batch(cfg, B, n_dyn=8, seed=1236)
    bs = BatchSolver(cfg)
    res = bs.solve(sc["p"]); t0 = bs.last_timing()["solve_ms"]
    res = bs.solve(sc["p"]); t0 = min(t0, bs.last_timing()["solve_ms"])
    n_psi, _ = bs.last_eval_counts(B)
    for name, key in (("inner iterations", res.num_inner_iterations), ("psi evaluations", n_psi)):
        order = np.argsort(-key.astype(np.int64), kind="stable")
        ps = np.ascontiguousarray(sc["p"][order])
        r2 = bs.solve(ps); t1 = bs.last_timing()["solve_ms"]
        r2 = bs.solve(ps); t1 = min(t1, bs.last_timing()["solve_ms"])
        assert np.array_equal(r2.solution, res.solution[order])
        print(f"B={B}: as generated {t0:.1f} ms = {B/t0*1e3:.0f}/s; longest first by {name}: {t1:.1f} ms = {B/t1*1e3:.0f}/s ({100*(t0/t1-1):+.1f} %)")
    # shortest first (worst case) for reference
    order = np.argsort(n_psi, kind="stable"); ps = np.ascontiguousarray(sc["p"][order])
    bs.solve(ps); t2 = bs.last_timing()["solve_ms"]
    print(f"B={B}: shortest first {t2:.1f} ms")
    bs.close()
