#!/usr/bin/env python3
"""Fuzz run (GPU box; test tooling -- it imports the oracle): random horizons, obstacle / robot counts MIXED inside one batch, scene
families, penalties and multipliers; the HIP path against the CPU oracle on psi / f / grad psi / F1 / F2 (1e-9 relative) and on short
tracked solves (same iteration counts, |du| small).  Also perturbs the tables the scenes never produce: rotated and time-varying
ellipses, rotated static polygons, terminal weights.  The two readings of the ALM penalty-stall rule (solver_penalty_stall) alternate
from trial to trial, on both sides.  usage: python tests/tools/fuzz_parity.py [trials = 40] [seed = 0] [full-solve trials = 0]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
from trajtrack_mpcndqn_rlboost_amd import MpcConfig, BatchSolver, scenes

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b)) / (1.0 + np.max(np.abs(b))))


def ocfg_of(cfg):
    return oracle.OracleConfig.from_dict(cfg.solver_dict())


worst = {"psi": 0.0, "grad": 0.0, "F1": 0.0, "F2": 0.0, "f": 0.0}
bad = []
t_start = time.time()
for trial in range(trials):
    rng = np.random.default_rng(seed0 * 1000 + trial)
    N = int(rng.choice([20, 40, 20, 40, int(rng.integers(8, 65))]))
    cfg = MpcConfig(N_hor=N)
    off = cfg.offsets()
    parts = []
    for sub in range(3):                                   # three sub-batches with different counts: mixed rows inside one launch
        n_dyn = int(rng.integers(0, cfg.Ndynobs + 1)); n_other = int(rng.integers(0, cfg.Nother + 1)) if hasattr(cfg, "Nother") else 0
        fam = str(rng.choice(["benchmark", "passing", "avoidance", "on_track"]))
        kw = dict(scenes.FAMILIES[fam])
        if n_dyn == 0: kw.pop("n_block", None)
        if "n_block" in kw and kw["n_block"][1] > n_dyn: kw["n_block"] = (1, max(1, n_dyn))
        if N < 12:
            kw = {}
        try:
            sc = scenes.make_batch(cfg, 16, n_dyn=n_dyn, n_other=n_other, seed=int(rng.integers(1 << 30)), **kw)
        except Exception as e:      # a family that does not fit this horizon: plain scenes
            sc = scenes.make_batch(cfg, 16, n_dyn=n_dyn, n_other=n_other, seed=int(rng.integers(1 << 30)))
        parts.append(sc["p"])
    p = np.concatenate(parts)
    B = p.shape[0]
    mode = int(rng.integers(0, 4))
    if mode == 1 and cfg.Ndynobs > 0:                      # rotated / time-varying ellipses in some rows (general tables)
        od = p[:, off["od"]:off["od"] + 6 * N * cfg.Ndynobs].reshape(B, cfg.Ndynobs, N, 6)
        act = od[..., 2] > 0
        od[..., 4] = np.where(act, rng.uniform(-1.5, 1.5, od[..., 4].shape), od[..., 4])
        if rng.random() < 0.5:
            od[..., 2] = np.where(act, od[..., 2] * rng.uniform(0.8, 1.3, od[..., 2].shape), od[..., 2])
    if mode == 2:                                          # terminal weights
        p[:, 8 + 5] = rng.uniform(0, 5, B); p[:, 8 + 6] = rng.uniform(0, 2, B)
    bs = BatchSolver(cfg)
    oc = ocfg_of(cfg)
    u = np.stack([rng.uniform(-0.7, 1.8, (B, N)), rng.uniform(-0.9, 0.9, (B, N))], axis=2).reshape(B, 2 * N)
    c = rng.choice([0.0, 10.0, 250.0, 6250.0, 1e6], B)
    y = rng.uniform(-3, 3, (B, 2 * N))
    r = bs.cost_grad(u, p, c, y)
    for i in range(B):
        o = oracle.cost_grad(oc, u[i], p[i], float(c[i]), y[i])
        for k in worst:
            e = rel(r[k][i], o[k])
            worst[k] = max(worst[k], e)
            if e > 1e-9:
                bad.append(("cost_grad", trial, N, i, k, e))
    bs.close()
    # short tracked solve: two outer iterations, every iteration count must agree
    stall = ("either", "both")[trial % 2]
    cfgk = MpcConfig(N_hor=N, solver_max_inner_iterations=int(rng.integers(3, 12)), solver_max_outer_iterations=2, solver_penalty_stall=stall)
    bs = BatchSolver(cfgk, latency_batch=int(rng.choice([0, 1 << 20])))
    u0 = np.tile([0.6, 0.1], (B, N)) + rng.uniform(-0.05, 0.05, (B, 2 * N))
    res = bs.solve(p, u0)
    uo, _, ro, _ = oracle.solve_batch(ocfg_of(cfgk), p, u0)
    same_it = np.array_equal(res.num_inner_iterations, ro["inner_iters"])
    du = np.max(np.abs(res.solution - uo), axis=1)
    # (a few steps from a cold Lipschitz estimate amplify rounding: single problems part by 1e-1, the median stays ~1e-9 .. 1e-7)
    if not same_it or np.median(du) > 1e-5 or not np.all(np.isfinite(res.solution)):
        bad.append(("solve", trial, N, int(np.argmax(du)), "du", float(du.max()), bool(same_it)))
    bs.close()
    # ... and FOUR outer iterations (round 6): the penalty-stall rule acts from the second outer iteration on, the two readings
    # alternate from trial to trial.  A problem that has parted by rounding may also take another exit: at least 90 % of the problems
    # must keep their iteration counts, the median distance stays small (N_hor = 40 amplifies rounding faster: DESIGN.md section 3).
    cfg4 = MpcConfig(N_hor=N, solver_max_inner_iterations=cfgk.solver_max_inner_iterations, solver_max_outer_iterations=4, solver_penalty_stall=stall)
    bs = BatchSolver(cfg4, latency_batch=int(rng.choice([0, 1 << 20])))
    res4 = bs.solve(p, u0)
    uo4, _, ro4, _ = oracle.solve_batch(ocfg_of(cfg4), p, u0)
    same4 = float(np.mean((res4.num_inner_iterations == ro4["inner_iters"]) & (res4.num_outer_iterations == ro4["outer_iters"])))
    du4 = np.max(np.abs(res4.solution - uo4), axis=1)
    # the yardstick for "small": what the oracle's own answer moves by when every parameter moves by one ulp (two float64 implementations
    # of this iteration cannot be closer than that; at N_hor = 40 it reaches 1e-2 after ~40 steps)
    us4, _, _, _ = oracle.solve_batch(ocfg_of(cfg4), np.nextafter(p, np.inf), u0)
    self4 = np.max(np.abs(us4 - uo4), axis=1)
    if same4 < 0.9 or np.median(du4) > max(1e-3, 10.0 * np.median(self4)) or not np.all(np.isfinite(res4.solution)):
        bad.append(("solve4", trial, N, int(np.argmax(du4)), "du", float(np.median(du4)), float(np.median(self4)), same4))
    shape = bs.last_shape()
    bs.close()
    print(f"trial {trial:3d} N {N:2d} mode {mode} stall {stall} latency {shape['latency_kernel']} max_dyn {shape['max_dyn']} max_static {shape['max_static']} "
          f"max_fleet {shape['max_fleet']}: cost/grad ok so far {not [b for b in bad if b[0] == 'cost_grad']}, solve du median {np.median(du):.1e} max {du.max():.1e} "
          f"same iteration counts {same_it}; four outer iterations: same counts {same4:.3f}, du median {np.median(du4):.1e} (oracle vs its 1-ulp twin {np.median(self4):.1e})", flush=True)

# ---- optional: FULL solves (reference caps) of mixed batches against the oracle: converged-on-both pairs within the north-star tolerance
n_full = int(sys.argv[3]) if len(sys.argv) > 3 else 0
both = agree = total = far = self_both = self_far = 0
worst_du = 0.0
for trial in range(n_full):
    rng = np.random.default_rng(seed0 * 1000 + 500 + trial)
    N = int(rng.choice([20, 20, 40]))
    stall = ("either", "both")[trial % 2]
    cfg = MpcConfig(N_hor=N, solver_penalty_stall=stall)
    parts = []
    for sub in range(3):
        n_dyn = int(rng.integers(1, cfg.Ndynobs + 1)); n_other = int(rng.integers(0, 4))
        fam = str(rng.choice(["passing", "avoidance", "on_track"]))
        kw = dict(scenes.FAMILIES[fam])
        if "n_block" in kw and kw["n_block"][1] > n_dyn: kw["n_block"] = (1, max(1, n_dyn))
        parts.append(scenes.make_batch(cfg, 8, n_dyn=n_dyn, n_other=n_other, seed=int(rng.integers(1 << 30)), **kw)["p"])
    p = np.concatenate(parts); B = p.shape[0]
    bs = BatchSolver(cfg, latency_batch=int(rng.choice([0, 1 << 20])))
    res = bs.solve(p)
    uo, _, ro, _ = oracle.solve_batch(ocfg_of(cfg), p, np.zeros((B, 2 * N)))
    bs.close()
    # the oracle against itself with every parameter moved by one ulp: what two float64 implementations may differ by
    u1, _, r1, _ = oracle.solve_batch(ocfg_of(cfg), np.nextafter(p, np.inf), np.zeros((B, 2 * N)))
    ok1 = (np.asarray(ro["status"]) == 0) & (np.asarray(r1["status"]) == 0)
    du1 = np.max(np.abs(u1 - uo), axis=1)
    self_both += int(ok1.sum()); self_far += int((ok1 & (du1 > 1e-3)).sum())
    so = np.asarray(ro["status"]); sg = res.status
    ok = (so == 0) & (sg == 0)
    du = np.max(np.abs(res.solution - uo), axis=1)
    both += int(ok.sum()); agree += int(((so == 0) == (sg == 0)).sum()); total += B
    if ok.any(): worst_du = max(worst_du, float(du[ok].max()))
    far += int((ok & (du > 1e-3)).sum())
    if ok.any() and du[ok].max() > 1e-3: bad.append(("full", trial, N, int(np.argmax(np.where(ok, du, 0))), float(du[ok].max())))
    print(f"full {trial:3d} N {N} stall {stall}: converged on both {int(ok.sum())}/{B}, max |du| on them {du[ok].max() if ok.any() else 0:.1e}, same converged/not {int(((so == 0) == (sg == 0)).sum())}/{B}", flush=True)
if n_full:
    print(f"full solves: {both} of {total} converged on both sides, {far} of them farther apart than 1e-3 (worst {worst_du:.2e}), agreement on which converge {agree / total:.3f}")
    print(f"oracle vs 1-ulp oracle on the same problems: {self_both} converged on both, {self_far} of them farther apart than 1e-3")
print("worst relative errors:", {k: f"{v:.2e}" for k, v in worst.items()})
print("failures:", bad if bad else "none", f"({time.time() - t_start:.0f} s)")
