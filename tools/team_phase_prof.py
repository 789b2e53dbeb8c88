#!/usr/bin/env python3
"""Where a PANOC step of the LATENCY kernel goes (needs a -DMPC_PROFILE build: MPCGPU_LIB=build_ab/libmpcgpu_prof.so): shader-clock cycles of
wavefront 0 of every team, by phase of the evaluation and by what happens between two passes (verdict barrier, adoption of the winning trial,
L-BFGS pair + direction, step residual, other logic).  usage: team_phase_prof.py [B = 8] [N_hor = 20]  -- B problems of the benchmark family,
one team each (B <= 2 x #CUs), so every wavefront has its SIMD to itself: the chain, not the throughput."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from trajtrack_mpcndqn_rlboost_amd import MpcConfig, BatchSolver, scenes
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
NH = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cfg = MpcConfig(N_hor=NH)
bs = BatchSolver(cfg)
sc = scenes.make_batch(cfg, B, n_dyn=8, seed=1236)
out = (C.c_double * 24)()
bs._L.mpcgpu_debug_read_prof.argtypes = [C.POINTER(C.c_double)]
res = bs.solve(sc["p"])
bs._L.mpcgpu_debug_read_prof(out)          # clear
res = bs.solve(sc["p"])
assert bs.last_shape()["latency_kernel"]
bs._L.mpcgpu_debug_read_prof(out)
t = np.array(out[:])
passes, steps = t[16], t[17]
tot = t[:16].sum() + t[18] + t[19]
names = ["headings", "positions+publish", "segments", "fleet+static", "dynamic", "pads+constraint sums", "phase B", "combine", "vector terms+psi", "adjoint"]
print(f"latency kernel, N_hor {NH}, {B} problems of the benchmark family (one team of four wavefronts each): solve {bs.last_timing()['solve_ms']:.1f} ms; per problem "
      f"{steps / B:.0f} PANOC steps, {passes / B:.0f} passes ({passes / steps:.2f} per step), {tot / steps:.0f} cycles per step (wavefront 0)")
ev = t[:10].sum()
print(f"  evaluation passes                {ev / steps:8.0f} cycles per step = {ev / passes:6.0f} per pass   {100 * ev / tot:5.1f} %")
for i, n in enumerate(names):
    print(f"      {n:26s} {t[i] / passes:8.0f} per pass")
for i, n in ((10, "verdict barrier (publish + wait for the slowest wavefront)"), (11, "adoption of the winning trial (LDS hand-over + barrier)"),
             (12, "L-BFGS pair update + direction (Gram form)"), (14, "step residual (two reductions, exit tests)"),
             (15, "end of a pass -> its verdict (Lipschitz / acceptance test, envelope sums)"), (18, "bookkeeping of a completed step up to the residual"),
             (19, "direction -> next pass (rhs, trial points, the loop's back edge + state dispatch)"), (13, "the rest (rare states)")):
    print(f"  {n:84s} {t[i] / steps:8.0f} cycles per step   {100 * t[i] / tot:5.1f} %")
