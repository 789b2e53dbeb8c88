#!/usr/bin/env python3
"""Looks at ONE full-solve case of tests/tools/fuzz_parity.py (seed, trial, problem): both answers, their costs and KKT figures.
usage: python tests/tools/fuzz_case.py seed trial problem"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
from support import kkt
from trajtrack_mpcndqn_rlboost_amd import MpcConfig, BatchSolver, scenes

seed0, trial, i = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(seed0 * 1000 + 500 + trial)
N = int(rng.choice([20, 20, 40]))
cfg = MpcConfig(N_hor=N)
parts, fams = [], []
for sub in range(3):
    n_dyn = int(rng.integers(1, cfg.Ndynobs + 1)); n_other = int(rng.integers(0, 4))
    fam = str(rng.choice(["passing", "avoidance", "on_track"]))
    kw = dict(scenes.FAMILIES[fam])
    if "n_block" in kw and kw["n_block"][1] > n_dyn: kw["n_block"] = (1, max(1, n_dyn))
    parts.append(scenes.make_batch(cfg, 8, n_dyn=n_dyn, n_other=n_other, seed=int(rng.integers(1 << 30)), **kw)["p"])
    fams.append((fam, n_dyn, n_other))
p = np.concatenate(parts); B = p.shape[0]
lat = int(rng.choice([0, 1 << 20]))
oc = oracle.OracleConfig.from_dict(cfg.solver_dict())
print("N", N, "families", fams, "problem", i, "->", fams[i // 8], "latency_batch", lat)
for lb in (0, 1 << 20):
    bs = BatchSolver(cfg, latency_batch=lb)
    res = bs.solve(p)
    print(f"GPU latency_batch={lb}: status {res.status[i]} inner {res.num_inner_iterations[i]} outer {res.num_outer_iterations[i]} cost {res.cost[i]:.12g} fpr {res.last_problem_norm_fpr[i]:.3e} f2 {res.f2_norm[i]:.3e}")
    bs.close()
    ug = res.solution[i]; yg = res.lagrange_multipliers[i]
uo, yo, ro, _ = oracle.solve_batch(oc, p, np.zeros((B, 2 * N)))
print(f"oracle: status {ro['status'][i]} inner {ro['inner_iters'][i]} outer {ro['outer_iters'][i]} cost {ro['cost'][i]:.12g}")
print("max |du|", float(np.abs(ug - uo[i]).max()), "at", int(np.argmax(np.abs(ug - uo[i]))))
for name, u, yy in (("gpu", ug, yg), ("oracle", uo[i], yo[i])):
    o = oracle.cost_grad(oc, u, p[i], 0.0, np.zeros(2 * N))
    chk = kkt.check_solution(cfg, oc, p[i], u, yy)
    print(name, "f", f"{o['f']:.12g}", {k: v for k, v in chk.items() if np.isscalar(v) or getattr(v, "shape", None) == ()})
# the 1-ulp oracle: the oracle against itself with every parameter moved by one ulp
p1 = np.nextafter(p, np.inf)
u1, _, r1, _ = oracle.solve_batch(oc, p1, np.zeros((B, 2 * N)))
print("oracle vs 1-ulp oracle: status", r1["status"][i], "max |du|", float(np.abs(u1[i] - uo[i]).max()))

# ---- where the GPU's and the oracle's iterations part (decision traces)
from trajtrack_mpcndqn_rlboost_amd.solver import variant_path
cap = 800
bs = BatchSolver(cfg, latency_batch=0, library=variant_path("trace"))
bs.set_trace(cap)
rt = bs.solve(p)
tg = bs.read_trace(B)[i]
bs.close()
_, rr, to, steps = oracle.solve_trace(oc, p[i], None, cap=cap)
n = min(int(rt.num_inner_iterations[i]), int(steps), cap)
F = oracle.TRACE_FIELDS
disc = [F.index(k) for k in F if k in ("alm_iteration", "iter", "lip_it", "lbfgs_active", "nls", "tau")]
first_disc = next((k for k in range(n) if not np.array_equal(tg[k, disc], to[k, disc])), None)
relerr = np.abs(tg[:n] - to[:n]) / (1e-300 + np.abs(to[:n]))
first_1e3 = next((k for k in range(n) if np.nanmax(relerr[k]) > 1e-3), None)
first_1e9 = next((k for k in range(n) if np.nanmax(relerr[k]) > 1e-9), None)
print(f"trace: GPU {int(rt.num_inner_iterations[i])} steps, oracle {int(steps)}; first step with a scalar off by > 1e-9: {first_1e9}, > 1e-3: {first_1e3}; first different discrete decision: {first_disc}")
for k in sorted(set(x for x in (first_1e9, first_1e3, first_disc) if x is not None)):
    print(" step", k, "gpu   ", dict(zip(F, np.round(tg[k], 12))))
    print(" step", k, "oracle", dict(zip(F, np.round(to[k], 12))))
