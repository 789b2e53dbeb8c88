#!/usr/bin/env python3
"""Summary of tools/env_counters.sh: env_step_kernel against the HBM roofline by the kernel-trace average, the counter traffic, and the
DYNAMIC VALU instruction count per environment step beside the static one of the edge loop (profiles/r05_env_isa_floor.txt).
usage: env_roofline.py <dir with kt_kernel_stats.csv, pmc_*_counter_collection.csv> [batch]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.roofline import _kernel_avg_ns, _read_counters, HBM_PEAK_GBS, NUM_CUS, SIMDS_PER_CU  # noqa: E402

KERNEL = "env_step_kernel"
ALGO_BYTES_PER_STEP = 7413       # map record read once + state + observations + reward + flag


def main():
    d = sys.argv[1]
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
    avg_ns, calls = _kernel_avg_ns(os.path.join(d, "kt_kernel_stats.csv"), KERNEL)
    sq, n_sq = _read_counters(os.path.join(d, "pmc_sq_counter_collection.csv"), KERNEL)
    fe, n_fe = _read_counters(os.path.join(d, "pmc_fetch_counter_collection.csv"), KERNEL)
    wr, n_wr = _read_counters(os.path.join(d, "pmc_write_counter_collection.csv"), KERNEL)
    sq = {k: v / n_sq for k, v in sq.items()}
    traffic = (fe["FETCH_SIZE"] / n_fe + wr["WRITE_SIZE"] / n_wr) * 1024.0
    algo = ALGO_BYTES_PER_STEP * B
    clock_ghz = sq["GRBM_GUI_ACTIVE"] / 8.0 / avg_ns
    quad = sq["GRBM_GUI_ACTIVE"] / 8.0 / 4.0 * NUM_CUS * SIMDS_PER_CU
    out = {"kernel": KERNEL, "batch": B, "kernel_avg_us_kernel_trace": avg_ns * 1e-3, "calls": calls,
           "env_steps_per_s_by_kernel_trace": B / (avg_ns * 1e-9),
           "algorithmic_bytes_per_launch": algo, "achieved_GBps": algo / avg_ns, "hbm_frac": algo / avg_ns / HBM_PEAK_GBS,
           "traffic_bytes_per_launch": traffic, "traffic_over_algorithmic": traffic / algo, "traffic_GBps": traffic / avg_ns,
           "valu_instructions_per_env_step_dynamic": sq["SQ_INSTS_VALU"] / B,
           "valu_busy_fraction": sq["SQ_ACTIVE_INST_VALU"] / quad, "resident_waves_per_simd": sq["SQ_WAVE_CYCLES"] / quad,
           "valu_issue_frac_of_peak": sq["SQ_INSTS_VALU"] / (avg_ns * 1e-9) / (NUM_CUS * SIMDS_PER_CU * clock_ghz * 1e9 / 4.0),
           "clock_GHz": clock_ghz}
    with open(os.path.join(d, "env_roofline.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
