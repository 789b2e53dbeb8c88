#!/usr/bin/env python3
"""Experiment: fill the tail of a throughput-kernel launch with latency-kernel workgroups on a LOW-priority stream.
The first B - T problems go to the throughput kernel on a high-priority stream, the last T to the latency kernel (four
wavefronts per problem, ~2x faster per problem, ~1.75x the work) on a low-priority stream launched at the same time; if the
dispatcher honours the priorities, the latency workgroups only get slots when the throughput kernel drains.
usage: tail_fill.py [B = 32768] [family]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from trajtrack_mpcndqn_rlboost_amd import MpcConfig, BatchSolver, scenes

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
fam = sys.argv[2] if len(sys.argv) > 2 else "bench"
kw = dict(dyn_clearance=0.1, box_clearance=0.3) if fam == "passing" else {}
cfg = MpcConfig()
dev = torch.device("cuda:0")
p = torch.from_numpy(scenes.make_batch(cfg, B, n_dyn=8, seed=1236, **kw)["p"]).to(dev)
lo_p, hi_p = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
print("stream priority range (low, high):", lo_p, hi_p)
main = BatchSolver(cfg, latency_batch=0)
tail = BatchSolver(cfg, latency_batch=1 << 20)
s_hi, s_lo = torch.cuda.Stream(priority=hi_p), torch.cuda.Stream(priority=lo_p)


def outs(n):
    return dict(u=torch.empty(n, 40, dtype=torch.float64, device=dev), cost=torch.empty(n, dtype=torch.float64, device=dev),
                status=torch.empty(n, dtype=torch.int32, device=dev))


ref = None
for T in (0, 256, 512, 1024, 2048, 4096):
    o1, o2 = outs(B - T), outs(max(T, 1))
    best = 1e30
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        main.solve_device(p[:B - T], o1, stream=s_hi.cuda_stream)
        if T:
            tail.solve_device(p[B - T:], o2, stream=s_lo.cuda_stream)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    u = torch.cat([o1["u"], o2["u"][:T]]).cpu().numpy()
    if ref is None:
        ref = u
    print(f"{fam} B={B} tail problems on the latency kernel {T:5d}: {1e3 * best:8.1f} ms = {B / best:8.0f} solves/s; bitwise equal to T = 0: {np.array_equal(u, ref)}")
