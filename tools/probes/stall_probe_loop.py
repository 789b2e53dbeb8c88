#!/usr/bin/env python3
"""The closed loop of tools/closed_loop.py (8192 robots, ordered) repeated: worst tick of every run -- a stall of the concurrent continuation
shows as a tick of seconds.  usage: python tools/probes/stall_probe_loop.py [runs = 10] [warm = 1] [robots = 8192]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tools.closed_loop import device_closed_loop
from trajtrack_mpcndqn_rlboost_amd import MpcConfig
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 10
warm = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
worst = []
for i in range(runs):
    r = device_closed_loop(MpcConfig(), B, 30, 5, 4, warm, "longest_first", seed=5 + i)
    worst.append(r["ms_per_tick_min_max"][1])
    print(f"run {i}: mean {r['ms_per_tick']:.1f} ms, worst tick {worst[-1]:.1f} ms", flush=True)
print(f"{os.environ.get('MPCGPU_LIB', 'libmpcgpu.so')}: {runs} runs x 30 ticks, worst tick overall {max(worst):.1f} ms, runs with a tick over 1 s: {sum(w > 1000 for w in worst)}")
