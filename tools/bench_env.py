#!/usr/bin/env python3
"""Secondary benchmark: environment steps per second of the fused HIP step kernel.

    python tools/bench_env.py [--batch B] [--steps K] [--warmup W] [--cpu-seconds S]

Prints ONE JSON line in the shape of bench.py's (metric / value / roofline / cpu_baseline).  A step = one
``env.step`` of B environments (two maps of tests/golden/env_rays_traces.npz, alternating), random actions, records
and state resident in HBM.  Algorithmic bytes per environment step: the map record (read once) + 2 x 256 B of state +
action + the two observation vectors + reward + flag.  The CPU leg is the numpy oracle (one core, bounded sample)."""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rl_env = importlib.import_module("trajtrack_mpcndqn_rlboost_amd.rl_env")
HBM_PEAK_GBS = 8000.0


def measured_traffic(B):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/r06_env_roofline.json, made by tools/env_counters.sh on
    this round's build), when they were collected on this batch size."""
    try:
        with open(os.path.join(ROOT, "profiles", "r06_env_roofline.json")) as fh:
            d = json.load(fh)
        return d["traffic_bytes_per_launch"] if d["batch"] == B else None
    except (OSError, KeyError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32768)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    args = ap.parse_args()
    fx = np.load(os.path.join(ROOT, "tests", "golden", "env_rays_traces.npz"))
    specs = json.loads(bytes(fx["specs_json"]).decode())
    maps = [rl_env.make_map(sp["boundary"], sp["static"], sp["dynamic"], sp["start"], sp["goal"], sp["path"])
            for sp in specs.values()]
    B = args.batch
    env = rl_env.BatchedRaysEnv([maps[i % len(maps)] for i in range(B)])
    env.reset()
    acts = torch.randint(0, 9, (args.warmup + args.steps, B), device=env.device, dtype=torch.int32)
    for t in range(args.warmup):
        env._launch(acts[t])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for t in range(args.steps):
        env._launch(acts[args.warmup + t])
    e1.record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    k_ms = e0.elapsed_time(e1) / args.steps
    rec_bytes = env.records.shape[1] * 8
    algo = rec_bytes + 2 * 8 * rl_env.STATE_DOUBLES + 4 + 4 * (rl_env.N_INTERNAL + rl_env.N_EXTERNAL) + 8 + 1
    achieved = algo * B / (k_ms * 1e-3) / 1e9
    line = {"metric": "DRL environment steps/sec (batch, rays + R1 reward)", "value": B * args.steps / elapsed,
            "unit": "env-steps/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{B} environments, maps scene1 (145 edges) / lhall alternating, random actions",
                       "batch_per_gpu": B, "record_bytes": rec_bytes},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(B), "kernel": "env_step_kernel",
                         "kernel_ms": k_ms, "algorithmic_bytes_per_env_step": algo}}
    if args.cpu_seconds > 0:
        from oracle import rl_env_numpy as orc
        o = orc.OracleRaysEnv(maps[0])
        rng = np.random.default_rng(0)
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < args.cpu_seconds:
            o.step(int(rng.integers(0, 9)))
            n += 1
        dt = time.perf_counter() - t0
        line["cpu_baseline"] = {"value": n / dt, "unit": "env-steps/s", "cores": 1, "kind": "port",
                                "sample": f"{n} steps of map scene1 in the numpy oracle, {dt:.1f} s"}
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
