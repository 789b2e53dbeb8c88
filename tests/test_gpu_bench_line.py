"""bench.py on the GPU at a small batch: the JSON line carries the contract's fields, `roofline` is consistent with the
run's own HIP-event kernel time, and the counter-based fields are measured in the run when rocprofv3 is there."""
import json
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def test_bench_line_contract_and_in_run_counters():
    B = 4096
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", str(B), "--steps", "2", "--warmup", "1",
                        "--cpu-seconds", "2", "--no-sweep", "--no-closed-loop"], cwd=ROOT, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                   # ONE JSON line
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "solves/s" and d["dtype"] == "f64" and d["n_gpus"] == 1 and d["steps"] == 2 and d["vs_baseline"] is None
    assert abs(d["value"] - B * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]
    ro = d["roofline"]
    assert ro["bound"] == "hbm" and ro["unit"] == "GB/s" and ro["peak"] == 8000.0
    assert ro["algorithmic_bytes_per_solve"] == 21944                      # 8 np + 16 N + 16 N + 40 at N = 20
    assert abs(ro["achieved"] - 21944 * B / (ro["kernel_ms"] * 1e-3) / 1e9) < 1e-9 * ro["achieved"] + 1e-12
    assert abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-15
    assert ro["kernel_ms"] <= d["ms_per_step"] * 1.001                     # the kernel fits into the step it is part of
    assert sum(d["config"]["status_histogram"]) == B
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0
    # the oracle's answers on its sample check the run they are timed beside: same problems converge, converged pairs within 1e-3
    par = cb["parity_on_sample"]
    assert par["problems"] >= cb["cores"] and par["tolerance"] == 1e-3 and par["same_converged_or_not"] >= 0.95
    if par["converged_on_both_sides"]:
        assert par["max_abs_du_on_them"] < 1e-3
    # ... and the tolerance rests on >= 256 converged pairs: the convergent families' samples (same horizon, same obstacle counts) feed it
    assert par["converged_pairs_total"] >= 256 and set(par["feeds"]) == {"headline_sample", "passing", "avoidance"}
    assert par["converged_pairs_total"] == sum(par["feeds"].values())
    assert par["pairs_beyond_tolerance_total"] <= 2                        # two local minima of a nonconvex problem: rare, reported
    assert par["max_abs_du_of_the_pairs_within_total"] < 1e-3
    # what `value` is made of, at the top level; the reading of the stall rule; the host-pointer boundary
    assert d["converged_fraction"] == d["config"]["converged_fraction"] and 0.0 <= d["converged_fraction"] <= 1.0
    assert abs(d["converged_solves_per_s"] - d["converged_fraction"] * d["value"]) < 1e-6 * d["value"] + 1e-9
    assert d["config"]["penalty_stall"] in ("either", "both")
    hb = d["config"]["host_boundary"]
    assert hb["pageable"]["value"] > 0 and hb["pinned"]["value"] > 0 and 0.0 <= hb["pinned"]["not_the_solve_kernels"] < 0.5
    assert hb["one_robot_per_call"]["calls"] >= 16 and hb["one_robot_per_call"]["ms_per_call_median"] > 0
    if ro["measured_in_run"]["traffic"]:
        assert shutil.which("rocprofv3") and ro["measured_in_run"]["secondary"]
        assert ro["traffic"] > 21944 * B                                   # the kernel's own cold state on top of the inputs
        sec = ro["secondary"]
        assert sec["measured_in_run"] and 0.1 < sec["valu_busy_frac_pmc"] <= 1.0
        assert 5e6 < sec["valu_instructions_per_solve"] < 3e7
    else:
        # no profiler on this box, or the counters were not available to this process: the line must say so (this batch has no
        # committed counter passes to fall back to, so the counter fields are null)
        print("counters not measured in this run:", ro.get("traffic_source"))
        assert ro["traffic"] is None and ro["secondary"] is None and not ro["measured_in_run"]["secondary"]


def test_two_ranks_share_the_one_gpu_through_the_real_solver():
    """The N > 1 path of bench.py with the REAL library: two rank processes (started by bench.py itself, gloo for the barrier and
    the reductions) shard the batch and share the one GPU of this box -- what the driver's 8-GPU run does with one GPU per rank.
    Each rank solves its own B robots (weak scaling); the line reports both ranks' rates and the whole-job value over the
    max-over-ranks time; side legs stay on the one-rank run."""
    B = 4096
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["MPCGPU_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", str(B), "--steps", "2", "--warmup", "1",
                        "--cpu-seconds", "0"], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["backend"] == "gloo" and d["data"] == "synthetic"
    assert len(d["per_rank_solves_per_s"]) == 2 and all(v > 0 for v in d["per_rank_solves_per_s"])
    assert abs(d["value"] - 2 * B * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]
    assert sum(d["config"]["status_histogram"]) == B          # rank 0's shard
    for k in ("convergent", "avoidance", "batch_sweep", "closed_loop"):
        assert k not in d["config"], k
    assert d["roofline"]["measured_in_run"]["traffic"] is False   # counter passes belong to the one-rank run
