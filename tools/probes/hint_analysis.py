#!/usr/bin/env python3
"""Offline: how good are hints for MPCGPU_OPT_ORDER?  Reads gpurun_out/hint_probe_<start>.npz (tools/probes/hint_probe.py) and
replays every tick through a fluid model of the launch -- 4096 resident wavefronts, all running at the same rate, the chip's
throughput a concave function of the number of resident wavefronts (calibrated on the measured as-given / perfect-hint times) --
for several predictors of a problem's psi-evaluation count.  usage: hint_analysis.py [cold|warm]"""
import heapq, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
start = sys.argv[1] if len(sys.argv) > 1 else "cold"
d = np.load(os.path.join(ROOT, "gpurun_out", f"hint_probe_{start}.npz"))
E, S, O, I, MS = d["evals"].astype(float), d["status"], d["outer"], d["inner"], d["ms"]
T, B = E.shape
SLOTS = 4096
# relative throughput of the chip with n resident wavefronts (per SIMD: 1, 2, 3, 4 wavefronts -> VALU share of the issue peak)
UW = np.array([0.0, 0.47, 0.74, 0.88, 0.936]) / 0.936
def thr(n):
    w = n / 1024.0
    i = min(int(w), 3)
    return (UW[i] + (UW[i + 1] - UW[i]) * (w - i)) * 1.0
def makespan(dur, order):
    """total time (in evaluation units at full throughput = SLOTS evaluations per unit) of the launch"""
    heap, P, t, nxt = [], 0.0, 0.0, 0
    n = len(order)
    while nxt < n and len(heap) < SLOTS:
        heapq.heappush(heap, P + dur[order[nxt]]); nxt += 1
    while heap:
        f = heapq.heappop(heap)
        k = len(heap) + 1
        rate = thr(k) * SLOTS / k          # progress per wavefront per unit time
        t += (f - P) / rate
        P = f
        if nxt < n:
            heapq.heappush(heap, P + dur[order[nxt]]); nxt += 1
    return t
def lpt(pred):
    return np.argsort(-pred, kind="stable")
rows = []
for t in range(6, T):
    dur = E[t]
    prev, prev2 = E[t - 1], E[t - 2]
    st, ou = S[t - 1], O[t - 1]
    preds = {
        "as given": None,
        "previous tick (product)": prev,
        "max of two ticks": np.maximum(prev, prev2),
        "mean of two ticks": 0.5 * (prev + prev2),
        "trend 2 prev - prev2": np.maximum(2 * prev - prev2, 0),
        "previous + not-converged bonus": prev + 5000.0 * (st != 0),
        "status, then outer count, then previous": (st != 0) * 1e6 + ou * 1e4 + prev * 1e-2,
        "perfect": dur,
    }
    ideal = dur.sum() / SLOTS
    r = {"ideal": ideal}
    for name, p in preds.items():
        r[name] = makespan(dur, np.arange(B) if p is None else lpt(p))
    rows.append(r)
names = list(rows[0].keys())
tot = {n: sum(r[n] for r in rows) for n in names}
print(f"{start} start, ticks 6..{T - 1}, model time relative to the ideal (all 4096 slots busy to the end):")
for n in names:
    print(f"  {n:45s} {tot[n] / tot['ideal']:.3f}   gain over as given {100 * (1 - tot[n] / tot['as given']):5.1f} %")
# how well does tick k predict tick k + 1?
cc = [np.corrcoef(E[t - 1], E[t])[0, 1] for t in range(6, T)]
print("correlation of evaluation counts between consecutive ticks: median %.2f (min %.2f)" % (np.median(cc), np.min(cc)))
conv = [(S[t] == 0).mean() for t in range(T)]
print("converged share per tick:", " ".join(f"{c:.2f}" for c in conv))
print("measured solve ms per tick:", " ".join(f"{m:.0f}" for m in MS))
# calibration check: model vs measured (as given), scaled by the eval rate of the last ticks
scale = np.median([MS[t] / makespan(E[t], np.arange(B)) for t in range(T - 8, T)])
print("model check (as given): measured / model*scale per tick:", " ".join(f"{MS[t] / (makespan(E[t], np.arange(B)) * scale):.2f}" for t in range(6, T, 3)))
