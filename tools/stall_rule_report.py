#!/usr/bin/env python3
"""profiles/r06_stall_rule.txt: what the two readings of the ALM penalty-stall rule do to every workload of the bench (GPU box).

The rule decides when the outer loop KEEPS the penalty c (yaml key solver_penalty_stall, option
MPCGPU_OPT_PENALTY_STALL; the reference configures the solver the rule belongs to at src/mpc_traj_tracker/mpc/mpc_generator.py:285-293):
    either  first outer iteration, or ||y+ - y|| OR ||F2|| shrank by theta   (the published engine as recalled; DEFAULT)
    both    first outer iteration, or both shrank                            (rounds 1-5)
For each family (benchmark / passing / avoidance at N_hor = 20, 8 discs; config 3 = N_hor = 40, 8 discs, B = 4096; the closed loop
of 8192 robots) and each rule: converged fraction, mean inner / outer iterations, solves/s (plain launches as given, HIP events),
the final penalty (the GPU path does not return it: the oracle's, on the first 256 problems, with the GPU-vs-oracle figures of that
sample beside it) and, between the rules, how far the answers move.

usage: python tools/stall_rule_report.py [--batch 32768] [--steps 3] > profiles/r06_stall_rule.txt"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle                                                           # noqa: E402  (report tooling: the oracle is the checker)
from trajtrack_mpcndqn_rlboost_amd import BatchSolver, MpcConfig, scenes  # noqa: E402

SAMPLE = 256


def run_family(N, family, B, seed, steps):
    out = {}
    p = None
    for stall in ("either", "both"):
        cfg = MpcConfig(N_hor=N, solver_penalty_stall=stall)
        if p is None:
            p = scenes.make_family(cfg, B, family, n_dyn=8, seed=seed)["p"]
        bs = BatchSolver(cfg, order="as_given")
        res = bs.solve(p)                       # warm-up (finds the shape)
        ms = []
        for _ in range(steps):
            res = bs.solve(p)
            ms.append(bs.last_timing()["solve_ms"])
        n_psi, _ = bs.last_eval_counts(B)
        bs.close()
        uo, _, ro, _ = oracle.solve_batch(oracle.OracleConfig.from_dict(cfg.solver_dict()), p[:SAMPLE])
        both = (res.status[:SAMPLE] == 0) & (ro["status"] == 0)
        du = np.max(np.abs(res.solution[:SAMPLE] - uo), axis=1)
        pen = np.asarray(ro["penalty"])
        out[stall] = dict(res=res, ms=float(np.mean(ms)), pen=pen, both=int(both.sum()), du=float(du[both].max()) if both.any() else float("nan"),
                          agree=float(np.mean((res.status[:SAMPLE] == 0) == (ro["status"] == 0))), n_psi=float(n_psi.mean()))
        r = out[stall]
        print(f"  {stall:6s}: converged {np.mean(res.status == 0):.4f}  mean inner {res.num_inner_iterations.mean():7.1f}  mean outer "
              f"{res.num_outer_iterations.mean():5.2f}  mean psi evaluations {r['n_psi']:8.1f}  median ||F2|| {np.median(res.f2_norm):.2e}  "
              f"solve kernel {r['ms']:8.1f} ms = {B / r['ms'] * 1e3:9.0f} solves/s ({np.sum(res.status == 0) / r['ms'] * 1e3:8.0f} converged/s)")
        vals, cnt = np.unique(pen, return_counts=True)
        print(f"          final penalty, oracle on the first {SAMPLE}: median {np.median(pen):.0f}; histogram "
              + ", ".join(f"{v:.0f}: {c}" for v, c in zip(vals, cnt)) + f"  | GPU vs oracle on that sample: converged on both {r['both']}, "
              f"max |du| there {r['du']:.2e}, same converged-or-not {r['agree']:.3f}")
    a, b = out["either"]["res"], out["both"]["res"]
    du = np.max(np.abs(a.solution - b.solution), axis=1)
    cc = (a.status == 0) & (b.status == 0)
    same = du == 0.0
    print(f"  between the rules: bitwise-identical answers {np.mean(same):.3f} of the batch; converged under both {int(cc.sum())}: of those identical "
          f"{int((cc & same).sum())}, |du|inf median {np.median(du[cc]) if cc.any() else float('nan'):.2e} p90 "
          f"{np.quantile(du[cc], 0.9) if cc.any() else float('nan'):.2e} max {du[cc].max() if cc.any() else float('nan'):.2e}; all problems: median "
          f"{np.median(du):.2e}; cost f(u) of the converged-under-both: mean either {a.cost[cc].mean() if cc.any() else float('nan'):.4f} both "
          f"{b.cost[cc].mean() if cc.any() else float('nan'):.4f}")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32768)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--no-closed-loop", action="store_true")
    args = ap.parse_args()
    t0 = time.time()
    print(__doc__.split("usage:")[0])
    for fam, seed in (("benchmark", 1234), ("passing", 4321), ("avoidance", 8642)):
        print(f"== N_hor = 20, 8 dynamic obstacles, family '{fam}' (scenes.FAMILIES), B = {args.batch}, cold start, {args.steps} timed launches")
        run_family(20, fam, args.batch, seed, args.steps)
    for fam, seed in (("benchmark", 1234), ("passing", 4321)):
        print(f"== config 3: mpc_longiter.yaml shape N_hor = 40, 8 dynamic obstacles, family '{fam}', B = 4096")
        run_family(40, fam, 4096, seed, args.steps)
    if not args.no_closed_loop:
        from tools.closed_loop import device_closed_loop
        print("== closed loop: DeviceTracker, 8192 robots, scene 1 with 4 constant-velocity discs, 30 timed ticks after 5 (tools/closed_loop.py), cold start, "
              "default order")
        for stall in ("either", "both"):
            cfg = MpcConfig(N_hor=20, solver_penalty_stall=stall)
            r = device_closed_loop(cfg, 8192, 30, 5, 4, False, "longest_first")
            per = r["status_histogram_per_tick"]
            print(f"  {stall:6s}: {r['ms_per_tick']:.1f} ms per tick (min / max {r['ms_per_tick_min_max'][0]:.1f} / {r['ms_per_tick_min_max'][1]:.1f}) = "
                  f"{r['value']:.0f} solves/s; converged {r['converged_fraction']:.3f}; mean inner {r['mean_inner_iterations']:.0f}; status histogram first tick "
                  f"{per[0]}, tick 10 {per[9]}, tick 20 {per[19]}, last tick {per[-1]}; entered the box {r['entered_box']:.4f}, touched a disc (0.8 m) "
                  f"{r['touched_disc']:.4f}, mean x after {r['mean_x_after']:.3f}")
    print(f"({time.time() - t0:.0f} s)")


if __name__ == "__main__":
    main()
