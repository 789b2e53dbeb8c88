#!/usr/bin/env python3
"""Offline: if the throughput launch knew, after the first k inner problems of every solve, how long the solve had been so far -- how
good a dispatch order would that give for the REST?  Fluid model of tools/probes/hint_analysis.py on gpurun_out/prefix_hint_<fam>_<B>.npz."""
import heapq, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fam = sys.argv[1] if len(sys.argv) > 1 else "bench"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
d = np.load(os.path.join(ROOT, "gpurun_out", f"prefix_hint_{fam}_{B}.npz"))
SLOTS = 4096
UW = np.array([0.0, 0.47, 0.74, 0.88, 0.936]) / 0.936
def thr(n):
    w = n / 1024.0; i = min(int(w), 3)
    return UW[i] + (UW[i + 1] - UW[i]) * (w - i)
def makespan(dur, order):
    heap, P, t, nxt, n = [], 0.0, 0.0, 0, len(order)
    while nxt < n and len(heap) < SLOTS:
        heapq.heappush(heap, P + dur[order[nxt]]); nxt += 1
    while heap:
        f = heapq.heappop(heap); k = len(heap) + 1
        t += (f - P) / (thr(k) * SLOTS / k); P = f
        if nxt < n:
            heapq.heappush(heap, P + dur[order[nxt]]); nxt += 1
    return t
total = d["evals_10"].astype(float)
ideal = total.sum() / SLOTS
asg = makespan(total, np.arange(B))
print(f"{fam}, B = {B}: ideal {ideal:.0f}, as given {asg / ideal:.3f} x ideal, perfect order {makespan(total, np.argsort(-total)) / ideal:.3f}")
for k in (1, 2, 3):
    pre = d[f"evals_{k}"].astype(float)
    rest = np.maximum(total - pre, 0.0)
    done = d[f"outer_10"] <= k            # solves that end within the first k outer iterations
    cc = np.corrcoef(pre, rest)[0, 1]
    # phase 1: everything runs its first k inner problems (as given); phase 2: the rest, longest-first by a predictor
    p1 = makespan(pre, np.arange(B))
    act = np.where(rest > 0)[0]
    preds = {"as given": None, "evaluations so far": pre[act], "inner iterations so far": d[f"inner_{k}"][act].astype(float),
             "not converged so far, then evaluations": (d[f"status_{k}"][act] != 0) * 1e6 + pre[act], "perfect": rest[act]}
    row = []
    for name, pr in preds.items():
        order = act if pr is None else act[np.argsort(-pr, kind="stable")]
        row.append(f"{name}: {(p1 + makespan(rest, order)) / ideal:.3f}")
    print(f"  split after {k} inner problem(s): corr(so far, rest) {cc:.2f}; phase 1 alone {p1 / ideal:.3f}; total = " + "; ".join(row))
