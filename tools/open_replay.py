#!/usr/bin/env python3
"""Pin the PANOC / ALM half of the oracle against a REAL OpEn build -- on a machine that has one.

Why this file exists.  The reference generates its solver on the user's machine (opengen 0.7.1 -> CasADi C code + the Rust crate
`optimization_engine` -> a PyO3 module; src/mpc_traj_tracker/mpc/mpc_generator.py:269-297, requirements.txt:1,26).  None of
those tools is in the image this repository is built in, so `oracle/mpc_oracle.c`'s solver iteration restates the PUBLISHED
algorithm and is "parity unpinned" (DESIGN.md, section 3): its answers are checked by scipy, its iteration only against itself.
This script is the missing link.  It never runs on the GPU box and imports nothing from the GPU package.

  step 1 (a machine WITH opengen 0.7.1, casadi 3.5.5, cargo and the reference checkout):
      python tools/open_replay.py record --reference /path/to/TrajTrack-MPCnDQN-RLBoost [--config mpc_default.yaml] [--build]
    builds the reference's own solver through the reference's own code (MpcModule(config).build(unicycle_model), exactly as
    src/test_block_mpc.py:33-36 does), loads it the way the reference does (trajectory_generator.py:63-71), feeds it every
    parameter vector the committed fixtures hold (tests/golden/*.npz: vectors the reference's own harness assembled) with the
    reference's call pattern `solver.run(p)` (initial_guess=None: cold start, src/interface_mpc.py:82), and writes
    open_replay_<optimizer_name>.npz: solution, cost, exit_status, num_outer_iterations, num_inner_iterations,
    last_problem_norm_fpr, f2_norm, penalty, solve_time_ms of every call.

  step 2 (any machine with this repository and a C compiler; no GPU, no OpEn):
      python tools/open_replay.py compare open_replay_navi_default.npz
    solves the same vectors with the oracle (`oracle.solve_batch`) under the 2 x 2 matrix of readings that cannot be checked here
    (line-search fallback last_trial | half_step  x  penalty stall rule either | both) and prints, per reading and fixture:
    agreement of the exit statuses, of the outer / inner iteration counts, of the FINAL PENALTY (the direct witness of the stall
    rule), max |u_open - u_oracle| over the calls that converge on both sides, and the first call whose iteration counts differ;
    the last line names the reading that matches the recording best.  Equal iteration counts on the short solves and
    |du| at rounding level is what "parity green" would mean; a systematic difference in the counts points at the constant or the
    rule in oracle/mpc_oracle.c that misreads the crate (DESIGN.md section 3 lists the readings that could not be checked).
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
# fixture file -> names of the arrays of parameter vectors in it (all assembled by the reference's own harness)
SOURCES = {"harness_traces.npz": ["A_p", "B_p"], "protocol_trace.npz": ["p"], "fleet_trace.npz": ["p"], "costgrad_N20.npz": ["p"]}
FIELDS = ("solution", "cost", "exit_status", "num_outer_iterations", "num_inner_iterations", "last_problem_norm_fpr", "f2_norm", "penalty",
          "solve_time_ms")
STATUS = {"Converged": 0, "NotConvergedIterations": 1, "NotConvergedOutOfTime": 2}


def parameter_vectors():
    """[(label, p [n, np])] of every committed fixture that holds whole parameter vectors (N_hor = 20 layout, np = 2658)."""
    out = []
    for fn, keys in SOURCES.items():
        d = np.load(os.path.join(GOLDEN, fn))
        for k in keys:
            p = np.asarray(d[k], dtype=np.float64)
            out.append((f"{fn}:{k}", p.reshape(-1, p.shape[-1])))
    return out


def record(args):
    ref_src = os.path.join(os.path.abspath(args.reference), "src")
    if not os.path.isdir(ref_src):
        sys.exit(f"{ref_src} not found: --reference must point at a checkout of Woodenonez/TrajTrack-MPCnDQN-RLBoost")
    sys.path.insert(0, ref_src)
    try:
        from util.mpc_config import Configurator                      # the reference's own modules: need casadi + opengen
        from pkg_motion_model import motion_model
        from mpc_traj_tracker.mpc.mpc_generator import MpcModule
    except ImportError as e:
        sys.exit(f"cannot import the reference's modules ({e}): this step needs opengen==0.7.1, casadi==3.5.5 and a Rust toolchain")
    cfg = Configurator(os.path.join(os.path.abspath(args.reference), "config", args.config))
    solver_dir = os.path.join(os.path.abspath(args.reference), cfg.build_directory, cfg.optimizer_name)
    if args.build or not os.path.isdir(solver_dir):
        cwd = os.getcwd()
        os.chdir(os.path.abspath(args.reference))                      # the builder writes to config.build_directory relative to the cwd
        try:
            MpcModule(cfg).build(motion_model.unicycle_model)
        finally:
            os.chdir(cwd)
    sys.path.append(solver_dir)                                        # trajectory_generator.py:65-71
    solver = __import__(cfg.optimizer_name).solver()
    rec = {f: [] for f in FIELDS}
    labels, counts, failures = [], [], 0
    for label, P in parameter_vectors():
        labels.append(label); counts.append(len(P))
        for p in P:
            s = solver.run(p=[float(v) for v in p])                    # initial_guess=None: cold start, as every reference call site
            if s is None:                                              # the binding returns None on an error (e.g. NotFiniteComputation)
                failures += 1
                for f in FIELDS:
                    rec[f].append(np.full(2 * int(cfg.N_hor), np.nan) if f == "solution" else (-1 if f == "exit_status" else np.nan))
                continue
            rec["solution"].append(np.asarray(s.solution, dtype=np.float64))
            rec["cost"].append(float(s.cost))
            rec["exit_status"].append(STATUS.get(str(s.exit_status), 9))
            for f in ("num_outer_iterations", "num_inner_iterations", "last_problem_norm_fpr", "f2_norm", "penalty", "solve_time_ms"):
                rec[f].append(float(getattr(s, f)))
    out = args.out or f"open_replay_{cfg.optimizer_name}.npz"
    import opengen
    np.savez_compressed(out, labels=np.array(labels), counts=np.array(counts), opengen_version=str(getattr(opengen, "__version__", "?")),
                        config=args.config, **{f: np.array(v) for f, v in rec.items()})
    print(f"wrote {out}: {sum(counts)} calls ({failures} returned None), statuses {np.bincount(np.maximum(np.array(rec['exit_status']), 0)).tolist()}")


def compare(args):
    sys.path.insert(0, ROOT)
    import oracle                                                      # builds oracle/_build/libmpc_oracle.so with gcc when missing
    sys.path.insert(0, os.path.join(ROOT, "trajtrack_mpcndqn_rlboost_amd"))
    import importlib.util
    spec = importlib.util.spec_from_file_location("mpc_config_only", os.path.join(ROOT, "trajtrack_mpcndqn_rlboost_amd", "config.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)          # the yaml surface only: no GPU library is loaded
    d = np.load(args.file)
    labels, counts = [str(x) for x in d["labels"]], [int(x) for x in d["counts"]]
    vectors = parameter_vectors()
    assert [l for l, _ in vectors] == labels and [len(p) for _, p in vectors] == counts, "the fixtures differ from the ones that were replayed"
    print(f"{args.file}: opengen {d['opengen_version']}, {d['config']}, {sum(counts)} calls")
    # The readings of the published algorithm that could not be checked against the crate where this repository was built
    # (DESIGN.md section 3): what follows a line search without acceptance x when the ALM loop keeps the penalty.  The recording
    # decides: the final PENALTY of every call is in it (a direct witness of the stall rule: 10 * 5^k), and so are the counts.
    scores = {}
    for fallback in ("last_trial", "half_step"):
        for stall in ("either", "both"):
            cfg = mod.MpcConfig(solver_linesearch_fallback=fallback, solver_penalty_stall=stall)
            ocfg = oracle.OracleConfig.from_dict(cfg.solver_dict())
            o = 0
            print(f"-- oracle reading: line search without acceptance = {fallback}, penalty stall rule = {stall}")
            tot = dict(n=0, status=0, inner=0, outer=0, penalty=0)
            for (label, P), n in zip(vectors, counts):
                u, _, res, _ = oracle.solve_batch(ocfg, P)
                sl = slice(o, o + n); o += n
                st_o, st_r = np.asarray(res["status"]), d["exit_status"][sl].astype(int)
                both = (st_o == 0) & (st_r == 0)
                du = np.max(np.abs(u - d["solution"][sl]), axis=1)
                inner_eq = np.asarray(res["inner_iters"]) == d["num_inner_iterations"][sl].astype(int)
                outer_eq = np.asarray(res["outer_iters"]) == d["num_outer_iterations"][sl].astype(int)
                pen_r = np.asarray(d["penalty"][sl], dtype=float)
                pen_eq = np.isclose(np.asarray(res["penalty"]), pen_r, rtol=1e-9) | ~np.isfinite(pen_r) | (pen_r == 0.0)
                first = int(np.argmin(inner_eq)) if not inner_eq.all() else -1
                tot["n"] += n; tot["status"] += int(np.sum(st_o == st_r)); tot["inner"] += int(inner_eq.sum())
                tot["outer"] += int(outer_eq.sum()); tot["penalty"] += int(pen_eq.sum())
                print(f"  {label:28s} n {n:4d}  same status {np.mean(st_o == st_r):.3f}  converged on both {int(both.sum()):4d}  "
                      f"max|du| there {du[both].max() if both.any() else float('nan'):.3e}  same outer count {outer_eq.mean():.3f}  "
                      f"same inner count {inner_eq.mean():.3f}  same final penalty {pen_eq.mean():.3f}"
                      + (f"  first different call {first}: inner {int(res['inner_iters'][first])} vs "
                         f"{int(d['num_inner_iterations'][sl][first])}" if first >= 0 else ""))
            scores[(fallback, stall)] = tot
    print("== summary over all calls (fraction equal to the recording)")
    for (fallback, stall), t in scores.items():
        print(f"   {fallback:10s} x {stall:6s}: status {t['status'] / t['n']:.3f}  outer count {t['outer'] / t['n']:.3f}  inner count "
              f"{t['inner'] / t['n']:.3f}  final penalty {t['penalty'] / t['n']:.3f}")
    best = max(scores, key=lambda k: (scores[k]["inner"] + scores[k]["outer"] + scores[k]["penalty"], scores[k]["status"]))
    print(f"== reading whose iteration counts and penalties match the recording best: solver_linesearch_fallback={best[0]}, "
          f"solver_penalty_stall={best[1]}")


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    sub = ap.add_subparsers(dest="cmd", required=True)
    r = sub.add_parser("record", help="on a machine with opengen + cargo: build the reference solver and replay the fixtures through it")
    r.add_argument("--reference", required=True, help="checkout of Woodenonez/TrajTrack-MPCnDQN-RLBoost")
    r.add_argument("--config", default="mpc_default.yaml")
    r.add_argument("--build", action="store_true", help="rebuild the solver even when its directory exists")
    r.add_argument("--out", default=None)
    c = sub.add_parser("compare", help="diff a recorded replay against oracle.solve_batch (no OpEn, no GPU needed)")
    c.add_argument("file")
    args = ap.parse_args()
    (record if args.cmd == "record" else compare)(args)


if __name__ == "__main__":
    main()
