#!/usr/bin/env python3
"""Tail promotion A/B (MPCGPU_OPT_TAIL_PROMOTION): kernel time of plain launches for several capacities K of the continuation
launch (0 = off), on the bench family, with the hash of the results (must be the same in every column).
usage: tail_ab.py [N_hor] [B,B,...] [K,K,...] [reps] [family: bench|passing|avoidance] [order: as_given|longest_first]
MPCGPU_LIB=<.so> selects the build (e.g. variants/libmpcgpu_yieldstep.so with MPCGPU_TAIL_POLL=<steps>); MPCGPU_TAIL_CONCURRENT=0 puts the
continuation behind the throughput launch instead of beside it (read by BatchSolver)."""
import hashlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from trajtrack_mpcndqn_rlboost_amd import MpcConfig, BatchSolver, scenes

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
Bs = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "8192,32768").split(",")]
Ks = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "0,256,512,1024").split(",")]
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
fam = sys.argv[5] if len(sys.argv) > 5 else "bench"
order = sys.argv[6] if len(sys.argv) > 6 else "as_given"
poll = int(os.environ["MPCGPU_TAIL_POLL"]) if os.environ.get("MPCGPU_TAIL_POLL") else None
waves = int(os.environ["MPCGPU_TAIL_WAVES"]) if os.environ.get("MPCGPU_TAIL_WAVES") else None
cfg = MpcConfig(N_hor=N)
print(f"# {os.environ.get('MPCGPU_LIB', 'libmpcgpu.so')}  N_hor={N} family={fam} order={order} poll={poll} waves={waves} concurrent={os.environ.get('MPCGPU_TAIL_CONCURRENT', 'default (1)')}")
for B in Bs:
    sc = scenes.make_batch(cfg, B, n_dyn=8, seed=1236) if fam == "bench" else scenes.make_family(cfg, B, fam, n_dyn=8, seed=1236)
    for K in Ks:
        bs = BatchSolver(cfg, order=order, tail_promotion=K)
        if poll or waves:
            bs.set_tail_promotion(K, poll, waves)
        ts, moved = [], 0
        for _ in range(reps + (1 if order == "longest_first" else 0)):
            res = bs.solve(sc["p"])
            ts.append(bs.last_timing()["solve_ms"])
            moved = bs.last_tail_promotion()[1]
        if order == "longest_first":
            ts = ts[1:]     # the first call has no hints
        h = hashlib.sha1()
        for a in (res.solution, res.cost, res.status, res.num_inner_iterations, res.num_outer_iterations, res.lagrange_multipliers):
            h.update(np.ascontiguousarray(a).tobytes())
        print(f"B={B:7d} K={K:5d}: kernel {min(ts):8.1f} ms (runs {' '.join('%.1f' % t for t in ts)}) = {B / min(ts) * 1e3:7.0f} solves/s, "
              f"{moved:4d} promoted, sha1 {h.hexdigest()[:12]}", flush=True)
        bs.close()
