#!/usr/bin/env python3
"""A/B of the two lane layouts of the solve kernel (MPCGPU_OPT_PAIRING): one vs two problems per wavefront.
usage: ab_pairing.py [B] [family: bench|passing] [N_hor]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from trajtrack_mpcndqn_rlboost_amd import MpcConfig, BatchSolver, scenes

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
fam = sys.argv[2] if len(sys.argv) > 2 else "bench"
N = int(sys.argv[3]) if len(sys.argv) > 3 else 20
cfg = MpcConfig(N_hor=N)
kw = dict(dyn_clearance=0.1, box_clearance=0.3) if fam == "passing" else {}
sc = scenes.make_batch(cfg, B, n_dyn=8, seed=1236, **kw)
out = {}
for pairing in (1, 2):
    bs = BatchSolver(cfg, pairing=pairing)
    best = 1e30
    for rep in range(3):
        res = bs.solve(sc["p"])
        best = min(best, bs.last_timing()["solve_ms"])
    sh = bs.last_shape()
    out[pairing] = res
    print(f"{fam} N={N} B={B} problems/wavefront {sh['problems_per_wavefront']} ({sh['waves_per_simd']} waves/SIMD, carve {sh['lds_bytes']} B): "
          f"solve {best:.1f} ms = {B / best * 1e3:.0f} solves/s; status {np.bincount(res.status, minlength=3).tolist()}, "
          f"mean inner {res.num_inner_iterations.mean():.0f}")
    bs.close()
a, b = out[1], out[2]
both = (a.status == 0) & (b.status == 0)
du = np.max(np.abs(a.solution - b.solution), axis=1)
print(f"converged in both layouts {both.sum()}/{B}: |du|inf max {du[both].max() if both.any() else float('nan'):.2e}; all: median {np.median(du):.2e}")
