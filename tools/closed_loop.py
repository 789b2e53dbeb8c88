#!/usr/bin/env python3
"""Closed-loop run of the batched tracker on the MI355X: B robots drive the reference's scene 1 (corridor, inflated
box on the path, src/pkg_dqn/utils/map.py:292-305) while discs cross it; one GPU solve per control tick.
usage: closed_loop.py [B] [ticks] [n_dyn] [warm]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from trajtrack_mpcndqn_rlboost_amd import BatchedTracker, MpcConfig
from trajtrack_mpcndqn_rlboost_amd.feeders import constant_velocity_prediction, DYN_OBS_SIZE

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
T = int(sys.argv[2]) if len(sys.argv) > 2 else 80
K = int(sys.argv[3]) if len(sys.argv) > 3 else 2
cfg = MpcConfig()
rng = np.random.default_rng(5)
walls = [[(0.0, 1.5), (0.0, 1.6), (9.0, 1.6), (9.0, 1.5)], [(0.0, 8.4), (0.0, 8.5), (9.0, 8.5), (9.0, 8.4)],
         [(11.0, 1.5), (11.0, 1.6), (16.0, 1.6), (16.0, 1.5)], [(11.0, 8.4), (11.0, 8.5), (16.0, 8.5), (16.0, 8.4)]]
inflate = lambda poly, m=0.8: [(min(x for x, _ in poly) - m, min(y for _, y in poly) - m), (max(x for x, _ in poly) + m, min(y for _, y in poly) - m),
                               (max(x for x, _ in poly) + m, max(y for _, y in poly) + m), (min(x for x, _ in poly) - m, max(y for _, y in poly) + m)]
box = [(7.5, 3.0), (7.5, 4.0), (8.5, 4.0), (8.5, 3.0)]
static = [inflate(w) for w in walls] + [inflate(box)]
WARM = len(sys.argv) > 4 and sys.argv[4] == "warm"
bt = BatchedTracker(cfg, B, warm_start=WARM)
y0 = rng.uniform(3.0, 4.0, B)
for i in range(B):
    bt.initialization(i, np.array([0.6, y0[i], 0.0]), np.array([15.4, 3.5, 0.0]),
                      [(0.6, y0[i]), (6.0, 5.6), (10.0, 5.6), (15.4, 3.5)], "work")
    bt.update_static_constraints(i, static)
# discs: start right of the box, move left/down slowly (different per robot world)
pos = np.stack([rng.uniform(10.0, 14.0, (B, K)), rng.uniform(4.5, 7.0, (B, K))], axis=-1)
vel = np.stack([rng.uniform(-0.12, -0.04, (B, K)), rng.uniform(-0.03, 0.03, (B, K))], axis=-1)
inside_box = np.zeros(B, bool); hit_disc = np.zeros(B, bool)
t0 = time.time(); solve_ms = []
for t in range(T):
    pred = constant_velocity_prediction(pos - vel, pos, steps=cfg.N_hor)         # [B, K, N, 6]
    bt.set_dynamic_constraints(pred)
    actions, _, cost = bt.step("work")
    solve_ms.append(bt.solver.last_timing()["solve_ms"])
    pos = pos + vel
    x, y = bt.states[:, 0], bt.states[:, 1]
    inside_box |= (x > 7.5) & (x < 8.5) & (y > 3.0) & (y < 4.0)
    hit_disc |= (np.hypot(pos[..., 0] - x[:, None], pos[..., 1] - y[:, None]) < 0.8).any(axis=1)   # physical radius 0.8
wall = time.time() - t0
d_goal = np.hypot(bt.states[:, 0] - 15.4, bt.states[:, 1] - 3.5)
print(f"B={B} ticks={T} discs={K} {'warm' if WARM else 'cold'} start: wall {wall:.1f}s  kernel {np.mean(solve_ms):.1f} ms/tick  ({B * T / (np.sum(solve_ms) * 1e-3):.0f} solves/s in-kernel)")
print(f"  progress: mean x {bt.states[:, 0].mean():.2f} m (start 0.6), within 0.5 m of goal: {(d_goal < 0.5).mean():.2f}, still active {bt.active.mean():.2f}")
print(f"  safety  : entered the (un-inflated) box {inside_box.mean():.3f}, touched a disc (0.8 m) {hit_disc.mean():.3f}")
print(f"  last tick status histogram {np.bincount(bt.last_result.status, minlength=4).tolist()} mean inner it {bt.last_result.num_inner_iterations.mean():.0f}")
