"""CPU: the plumbing of tools/open_replay.py (the script that would pin the oracle's PANOC / ALM half against a real OpEn build).
`compare` is fed a recording made from the oracle itself, in the format `record` writes: every agreement figure must be 1."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_compare_accepts_a_recording_in_the_record_format(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import open_replay
    import oracle
    from trajtrack_mpcndqn_rlboost_amd.config import MpcConfig
    vectors = open_replay.parameter_vectors()
    assert sum(len(p) for _, p in vectors) >= 300 and all(p.shape[1] == 2658 for _, p in vectors)
    ocfg = oracle.OracleConfig.from_dict(MpcConfig().solver_dict())
    rec = {f: [] for f in open_replay.FIELDS}
    for _, P in vectors:
        u, _, res, _ = oracle.solve_batch(ocfg, P)
        rec["solution"].append(u); rec["exit_status"].append(res["status"])
        rec["num_outer_iterations"].append(res["outer_iters"]); rec["num_inner_iterations"].append(res["inner_iters"])
        rec["penalty"].append(res["penalty"])
        for f in ("cost", "last_problem_norm_fpr", "f2_norm", "solve_time_ms"):
            rec[f].append(np.zeros(len(P)))
    f = str(tmp_path / "open_replay_selftest.npz")
    np.savez_compressed(f, labels=np.array([l for l, _ in vectors]), counts=np.array([len(p) for _, p in vectors]), opengen_version="self-test",
                        config="mpc_default.yaml", **{k: np.concatenate(v) for k, v in rec.items()})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "open_replay.py"), "compare", f], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = [l for l in r.stdout.splitlines() if ".npz:" in l and "same status" in l]
    assert len(rows) == 4 * len(vectors)                     # 2 x 2 readings: line-search fallback x penalty stall rule
    first_reading = rows[:len(vectors)]                      # last_trial x either = the defaults the recording was made with
    for row in first_reading:
        assert "same status 1.000" in row and "same inner count 1.000" in row and "same outer count 1.000" in row, row
        assert "same final penalty 1.000" in row, row
    # ... and the tool names that reading as the one that matches
    assert "solver_linesearch_fallback=last_trial, solver_penalty_stall=either" in r.stdout.splitlines()[-1], r.stdout[-500:]
    # the other stall rule is told apart by the recording (the fixtures hold calls whose penalty grows under one rule only)
    summary = [l for l in r.stdout.splitlines() if " x both" in l and "final penalty" in l]
    assert summary and all("final penalty 1.000" not in l for l in summary), summary
    # the record step refuses politely without the reference's tools
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "open_replay.py"), "record", "--reference", str(tmp_path)], capture_output=True, text=True)
    assert r.returncode != 0 and "not found" in r.stderr
