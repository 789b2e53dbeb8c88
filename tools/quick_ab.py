#!/usr/bin/env python3
"""Kernel time and a hash of the results for one build (MPCGPU_LIB=<.so> selects it): the A/B tool of a kernel experiment.
usage: quick_ab.py [B] [reps] [N_hor] [family: bench|passing]
Prints the best solve-kernel time of `reps` launches and the SHA-1 of (solution, cost, status, iteration counts): two builds
that are meant to compute the same bits must print the same hash."""
import hashlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from trajtrack_mpcndqn_rlboost_amd import MpcConfig, BatchSolver, scenes

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
N = int(sys.argv[3]) if len(sys.argv) > 3 else 20
fam = sys.argv[4] if len(sys.argv) > 4 else "bench"
cfg = MpcConfig(N_hor=N)
kw = dict(dyn_clearance=0.1, box_clearance=0.3) if fam == "passing" else {}
sc = scenes.make_batch(cfg, B, n_dyn=8, seed=1236, **kw)
bs = BatchSolver(cfg)
ts = []
for _ in range(reps):
    res = bs.solve(sc["p"])
    ts.append(bs.last_timing()["solve_ms"])
h = hashlib.sha1()
for a in (res.solution, res.cost, res.status, res.num_inner_iterations, res.num_outer_iterations):
    h.update(np.ascontiguousarray(a).tobytes())
sh = bs.last_shape()
print(f"{os.environ.get('MPCGPU_LIB', 'libmpcgpu.so'):40s} N={N} B={B} {fam}: kernel {min(ts):8.1f} ms (runs {' '.join('%.1f' % t for t in ts)}) = "
      f"{B / min(ts) * 1e3:7.0f} solves/s  carve {sh['lds_bytes']} B, {sh['waves_per_simd']} waves/SIMD  sha1 {h.hexdigest()[:12]}")
