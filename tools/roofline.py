#!/usr/bin/env python3
"""Roofline numbers of the solve kernel, rebuilt from the raw rocprofv3 CSVs kept under profiles/.

    python tools/roofline.py rebuild  [--raw profiles/raw_r06] [--out profiles/r06_roofline_bench.json]
    python tools/roofline.py show     [profiles/r06_roofline_bench.json] [bench-line.json]

`rebuild` reads, from the raw directory (copies of what `tools/collect_profiles.sh` wrote on the GPU box):
    workload.json                 {"N_hor", "n_dyn", "batch_per_gpu"}: the bench.py arguments of every pass
    kt_kernel_stats.csv           rocprofv3 --kernel-trace --stats            (average kernel duration)
    pmc_sq_counter_collection.csv rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_SALU
                                              SQ_WAVES GRBM_GUI_ACTIVE
    pmc_fetch_counter_collection.csv   rocprofv3 --pmc FETCH_SIZE   (own pass: 3 of the 4 TCC slots)
    pmc_write_counter_collection.csv   rocprofv3 --pmc WRITE_SIZE
and writes the derived figures bench.py prints (bench.py uses the file only when its workload matches the run).

Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* count quad-cycles; GRBM_GUI_ACTIVE is summed over the 8
XCDs; FETCH_SIZE / WRITE_SIZE are KiB of L2 <-> fabric requests (Infinity-Cache hits included).  The guide's x2 correction
of FETCH_SIZE applies to 16-byte-per-lane streaming reads; this code base loads 8 bytes per lane, so the raw value is
calibrated instead on prep_kernel, which reads `p` exactly once (B x np x 8 bytes): the measured / expected ratio is
stored next to the raw counters and NO correction is applied when it is within 15 % of 1.

The flop table (`flops_per_eval`) counts the double-precision additions, multiplications and FMAs (2 flop) the device
code of one psi (and psi + grad psi) evaluation executes on ACTIVE lanes, by horizon and active rows; it is a static
count of mpc_kernels.hpp::eval_point, not a counter.
"""
from __future__ import annotations

import csv
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

NUM_CUS, SIMDS_PER_CU = 256, 4
SOLVE_KERNEL = "solve_kernel_pair"   # the dominant kernel (one problem per wavefront); "solve_kernel" alone would also match ...
TAIL_KERNEL = "solve_kernel_team"    # ... the latency kernel, which finishes the last problems of a launch since round 5
HBM_PEAK_GBS = 8000.0
FP64_VECTOR_PEAK_TF = 78.6


# ---------------------------------------------------------------------------------------------------------
# static f64 flop count of one evaluation (mpc_kernels.hpp::eval_point), per horizon step unless noted
# ---------------------------------------------------------------------------------------------------------
def flops_per_eval(N: int, Ks: int, Kf: int, Kd: int, grad: bool) -> float:
    """f64 flops of one psi evaluation (grad=False) or psi + grad psi (grad=True) at horizon N with Ks / Kf / Kd active
    static / fleet / dynamic rows (no violated hard constraint: item phase B adds ~25 per violated (row, step))."""
    rows_v = (N + 15) // 16
    per_step = 0.0
    # rollout: half-angle polynomials (2 x 6 Horner FMAs + 6), e^{i ts w} (5), complex prefix product
    # (4 + rows_v - 1 steps x 6), heading rotations (4 x 3), Simpson sums and derivatives (4 x 4),
    # position increments + 2 prefix sums (2 x (2 + 4 + rows_v - 1))
    per_step += 30 + 5 + (4 + rows_v - 1) * 6 + 12 + 16 + 2 * (6 + rows_v - 1)
    # reference segments: SEG_WIN * LPS of them are always evaluated (fewer near the end of the horizon: N - k remain);
    # 20 flops each (projection 6, clamp 2, offset 6, distance 3, compare) + 8 for the gradient of the running minimum
    lps = max(1, 64 // N)
    seg = sum(min(1 * lps, N - k) for k in range(N)) / N   # SEG_WIN = 1 (2 in rounds 1-3)
    per_step += seg * 24 + 6            # + pruning test against the suffix bounding circle
    per_step += Ks * 27                 # 4 half-planes (4 x 4), squares (4), products (3), compare; gradient only inside
    per_step += Kf * 6                  # squared distance + hinge
    per_step += Kd * 19                 # frame coordinates (8), two indicators (8), weight (1), compares
    per_step += 10                      # zero-padded rows in closed form
    per_step += 30 + 3 + 3 * lps        # per-step terms + F1 box distance, psi partial, combine of the item lanes
    if grad:
        # acceleration adjoint (10), input terms (8), three suffix scans (3 x (4 + rows_v - 1)), adjoint products (17)
        per_step += 10 + 8 + 3 * (4 + rows_v - 1) + 17
    return per_step * N


def flops_per_solve_kernel_launch(N, Ks, Kf, Kd, n_psi, n_grad) -> float:
    """Executed f64 flops of one launch from the per-problem evaluation counters (n_grad of the n_psi evaluations also
    produced the gradient).  The solver algebra between evaluations (~10 N-vector operations and the
    L-BFGS step) is added per PANOC iteration / gradient evaluation."""
    import numpy as np
    n_psi = np.asarray(n_psi, dtype=np.float64)
    n_grad = np.asarray(n_grad, dtype=np.float64)
    f_psi = flops_per_eval(N, Ks, Kf, Kd, False)
    f_grad = flops_per_eval(N, Ks, Kf, Kd, True)
    n_iter = n_psi - n_grad                       # one gradient-free evaluation (Lipschitz check) per PANOC iteration
    # L-BFGS step per PANOC iteration.  Gram form (N_hor = 20 and 40, round 3): pass 1 = 2 mem rows x 2N x 2 products, the two scalar
    # recurrences 2 x mem x (2 mem rows), pass 2 = 2 mem rows x 2N; two-loop form: 4 x mem dot products / updates of length 2N.
    mem = 10
    lbfgs = (2 * (2 * mem) * (2 * N) * 2 + 2 * (2 * mem) * (2 * N) + 2 * 2 * mem * 2 * mem) if N in (20, 40) else 2 * (4 * mem * 2 * N)
    algebra = n_iter * (lbfgs + 20 * 2 * N) + n_grad * (8 * 2 * N)
    return float(((n_psi - n_grad) * f_psi + n_grad * f_grad + algebra).sum())


# ---------------------------------------------------------------------------------------------------------
# raw CSV -> derived json
# ---------------------------------------------------------------------------------------------------------
def _read_counters(path, kernel_substr):
    """{counter: sum over all rows of the kernel}, number of dispatches of the kernel."""
    tot, disp = defaultdict(float), set()
    with open(path, newline="") as fh:
        for row in csv.DictReader(fh):
            if kernel_substr not in row["Kernel_Name"]:
                continue
            tot[row["Counter_Name"]] += float(row["Counter_Value"])
            disp.add(row["Dispatch_Id"])
    return dict(tot), len(disp)


def _dispatch_avg_ns(path, kernel_substr):
    """Average duration of the kernel's dispatches from the timestamps of a counter-collection CSV (every row of a dispatch
    carries them), and the number of dispatches."""
    dur = {}
    with open(path, newline="") as fh:
        for row in csv.DictReader(fh):
            if kernel_substr in row["Kernel_Name"] and row.get("Start_Timestamp") and row.get("End_Timestamp"):
                dur[row["Dispatch_Id"]] = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
    if not dur:
        raise KeyError(kernel_substr)
    return sum(dur.values()) / len(dur), len(dur)


def _kernel_avg_ns(path, kernel_substr):
    with open(path, newline="") as fh:
        for row in csv.DictReader(fh):
            if kernel_substr in row["Name"]:
                return float(row["AverageNs"]), int(row["Calls"])
    raise KeyError(kernel_substr)


def rebuild(raw_dir, out_path):
    with open(os.path.join(raw_dir, "workload.json")) as fh:
        wl = json.load(fh)
    B, N = wl["batch_per_gpu"], wl["N_hor"]
    np_ = 18 + 4 * N + 3 * N * 10 + 120 + 15 * 6 * N + 2 * N       # mpc_default.yaml dimensions
    algo = 8 * np_ + 16 * N * 2 + 40
    kt = os.path.join(raw_dir, "kt_kernel_stats.csv")
    if os.path.exists(kt):
        avg_ns, calls = _kernel_avg_ns(kt, SOLVE_KERNEL)
        prep_ns, _ = _kernel_avg_ns(kt, "prep_kernel")
        time_source = "rocprofv3 --kernel-trace --stats pass"
    else:   # bench.py's in-run passes: no kernel-trace pass, the dispatch timestamps of the SQ counter pass itself
        avg_ns, calls = _dispatch_avg_ns(os.path.join(raw_dir, "pmc_sq_counter_collection.csv"), SOLVE_KERNEL)
        prep_ns, _ = _dispatch_avg_ns(os.path.join(raw_dir, "pmc_sq_counter_collection.csv"), "prep_kernel")
        time_source = "dispatch timestamps of the SQ counter pass"
    sq, n_sq = _read_counters(os.path.join(raw_dir, "pmc_sq_counter_collection.csv"), SOLVE_KERNEL)
    fe, n_fe = _read_counters(os.path.join(raw_dir, "pmc_fetch_counter_collection.csv"), SOLVE_KERNEL)
    wr, n_wr = _read_counters(os.path.join(raw_dir, "pmc_write_counter_collection.csv"), SOLVE_KERNEL)
    # the continuation launch of the tail promotion (round 5): the latency kernel on the problems that left the throughput launch
    try:
        tail_ns, tail_calls = (_kernel_avg_ns(kt, TAIL_KERNEL) if os.path.exists(kt) else
                               _dispatch_avg_ns(os.path.join(raw_dir, "pmc_sq_counter_collection.csv"), TAIL_KERNEL))
    except KeyError:
        tail_ns, tail_calls = 0.0, 0
    pf, n_pf = _read_counters(os.path.join(raw_dir, "pmc_fetch_counter_collection.csv"), "prep_kernel")
    sq = {k: v / n_sq for k, v in sq.items()}
    fetch_kib, write_kib = fe["FETCH_SIZE"] / n_fe, wr["WRITE_SIZE"] / n_wr
    prep_fetch = pf["FETCH_SIZE"] / n_pf * 1024.0
    calib = prep_fetch / (B * np_ * 8.0)
    # clock from GRBM_GUI_ACTIVE (cycles the GPU was busy, summed over 8 XCDs) and the kernel duration of the SAME pass
    # is not available per pass; the kernel-trace duration of the same workload is used
    clock_ghz = sq["GRBM_GUI_ACTIVE"] / 8.0 / avg_ns
    simd_quad_cycles = sq["GRBM_GUI_ACTIVE"] / 8.0 / 4.0 * NUM_CUS * SIMDS_PER_CU
    derived = {
        "workload": wl,
        "kernel": wl.get("kernel", "solve_kernel_pair"),
        "kernel_avg_ms_kernel_trace": avg_ns * 1e-6, "kernel_calls_kernel_trace": calls, "kernel_time_source": time_source,
        "prep_kernel_avg_ms": prep_ns * 1e-6,
        # two launches of it per solve call when the continuation runs while the throughput launch drains (side stream + sweep):
        # per solve call = total duration / calls of the dominant kernel (the side-stream launch OVERLAPS the end of the dominant one)
        "tail_kernel": TAIL_KERNEL, "tail_kernel_avg_ms": tail_ns * 1e-6, "tail_kernel_calls": tail_calls,
        "tail_kernel_ms_per_solve_call": tail_ns * 1e-6 * tail_calls / max(calls, 1),
        "raw_per_launch": {"sq_pass": sq, "FETCH_SIZE_KiB": fetch_kib, "WRITE_SIZE_KiB": write_kib,
                           "prep_kernel_FETCH_SIZE_KiB": prep_fetch / 1024.0},
        "fetch_calibration_prep_kernel_measured_over_expected": calib,
        "fetch_correction_applied": 1.0,
        "algorithmic_bytes_per_launch": algo * B,
        "traffic_bytes_per_launch": (fetch_kib + write_kib) * 1024.0,
        "wasted_traffic_ratio": (fetch_kib + write_kib) * 1024.0 / (algo * B),
        "traffic_GBps": (fetch_kib + write_kib) * 1024.0 / avg_ns,
        "hbm_frac_algorithmic": algo * B / avg_ns / HBM_PEAK_GBS,
        "clock_GHz": clock_ghz,
        "valu_instructions_per_launch": sq["SQ_INSTS_VALU"],
        "salu_instructions_per_launch": sq.get("SQ_INSTS_SALU"),
        "valu_busy_fraction": sq["SQ_ACTIVE_INST_VALU"] / simd_quad_cycles,
        "resident_waves_per_simd": sq["SQ_WAVE_CYCLES"] / simd_quad_cycles,
        "valu_issue_frac_of_peak": sq["SQ_INSTS_VALU"] / (avg_ns * 1e-9) / (NUM_CUS * SIMDS_PER_CU * clock_ghz * 1e9 / 4.0),
    }
    with open(out_path, "w") as fh:
        json.dump(derived, fh, indent=1)
    return derived


def load_pmc_for(path, N, n_dyn, B):
    """Derived PMC figures -- only when they were collected on exactly this workload (else None)."""
    try:
        with open(path) as fh:
            d = json.load(fh)
        w = d["workload"]
        if (w["N_hor"], w["n_dyn"], w["batch_per_gpu"]) != (N, n_dyn, B):
            return None
        return d
    except (OSError, KeyError, ValueError):
        return None


def show(derived_path, bench_path=None):
    with open(derived_path) as fh:
        d = json.load(fh)
    B = d["workload"]["batch_per_gpu"]
    print(f"workload: {d['workload']}")
    print(f"solve kernel: {d['kernel_avg_ms_kernel_trace']:.1f} ms per launch (kernel trace, {d['kernel_calls_kernel_trace']} calls)")
    print(f"HBM roofline (algorithmic bytes): {d['algorithmic_bytes_per_launch'] / 1e6:.0f} MB -> frac {d['hbm_frac_algorithmic']:.2e}")
    print(f"measured L2<->fabric traffic: {d['traffic_bytes_per_launch'] / 1e9:.1f} GB per launch = {d['traffic_GBps']:.0f} GB/s "
          f"({100 * d['traffic_GBps'] / HBM_PEAK_GBS:.1f} % of peak), wasted ratio {d['wasted_traffic_ratio']:.0f}x; "
          f"FETCH calibration on prep_kernel {d['fetch_calibration_prep_kernel_measured_over_expected']:.3f}")
    print(f"VALU: {d['valu_instructions_per_launch'] / B:.3g} instructions per solve, busy {d['valu_busy_fraction']:.3f}, "
          f"{d['resident_waves_per_simd']:.2f} resident wavefronts/SIMD, clock {d['clock_GHz']:.2f} GHz, "
          f"issue fraction of peak {d['valu_issue_frac_of_peak']:.3f}")
    if bench_path:
        with open(bench_path) as fh:
            line = json.loads(fh.read().strip().splitlines()[-1])
        r = line["roofline"]
        print(f"bench line: {line['value']:.0f} solves/s, kernel {r['kernel_ms']:.1f} ms, hbm frac {r['frac']:.2e}, "
              f"flops {r['flops']['achieved']:.2f} TF ({100 * r['flops']['frac']:.1f} %), "
              f"valu issue {r['secondary']['frac'] if r.get('secondary') else None}")


if __name__ == "__main__":
    cmd = sys.argv[1] if len(sys.argv) > 1 else "show"
    if cmd == "rebuild":
        raw = os.path.join(ROOT, "profiles", "raw_r06")
        out = os.path.join(ROOT, "profiles", "r06_roofline_bench.json")
        a = sys.argv[2:]
        while a:
            if a[0] == "--raw": raw = a[1]
            elif a[0] == "--out": out = a[1]
            a = a[2:]
        d = rebuild(raw, out)
        print(json.dumps({k: v for k, v in d.items() if k != "raw_per_launch"}, indent=1))
    else:
        show(sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "r06_roofline_bench.json"),
             sys.argv[3] if len(sys.argv) > 3 else None)
