#!/usr/bin/env python3
"""Looks for stalls of the concurrent continuation: many solves of batches that finish almost at once (the throughput launch is then
limited by the rate at which workgroups are dispatched, and its last workgroups are still waiting when the launch starts to promote),
kernel time of every call; prints the distribution and the slowest calls.
usage: python tools/probes/stall_probe.py [calls = 300] [B = 8192] [family = on_track] [order = longest_first]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from trajtrack_mpcndqn_rlboost_amd import MpcConfig, BatchSolver, scenes

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
fam = sys.argv[3] if len(sys.argv) > 3 else "on_track"
order = sys.argv[4] if len(sys.argv) > 4 else "longest_first"
cfg = MpcConfig(N_hor=20)
sc = scenes.make_family(cfg, B, fam, n_dyn=4, seed=11)
bs = BatchSolver(cfg, latency_batch=0, order=order)
ts, moved = [], []
for i in range(calls):
    bs.solve(sc["p"])
    ts.append(bs.last_timing()["solve_ms"]); moved.append(bs.last_tail_promotion()[1])
ts = np.array(ts)
print(f"{os.environ.get('MPCGPU_LIB', 'libmpcgpu.so')} {fam} B={B} order={order}: {calls} calls, kernel ms min {ts.min():.2f} median {np.median(ts):.2f} "
      f"p99 {np.percentile(ts, 99):.2f} max {ts.max():.2f}; calls over 10 x median: {int(np.sum(ts > 10 * np.median(ts)))}; promoted per call {np.mean(moved):.0f}")
