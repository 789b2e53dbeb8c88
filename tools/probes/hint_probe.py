#!/usr/bin/env python3
"""What does tick k know about the length of tick k + 1?  Runs the device closed loop of tools/closed_loop.py (8192 robots, scene 1)
and records, per tick and robot: psi evaluations, status, outer and inner iterations (read back OUTSIDE any timing).  The file
gpurun_out/hint_probe_<start>.npz is analysed offline by tools/probes/hint_analysis.py (list-scheduling model of the launch).
usage: hint_probe.py [B] [ticks] [warm]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from trajtrack_mpcndqn_rlboost_amd import BatchSolver, MpcConfig
from trajtrack_mpcndqn_rlboost_amd.device_tracker import DeviceTracker
from trajtrack_mpcndqn_rlboost_amd.feeders import DYN_OBS_SIZE
from tools.closed_loop import scene_one, setup

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 35
warm = len(sys.argv) > 3 and sys.argv[3] == "warm"
cfg = MpcConfig()
N, n_dyn = int(cfg.N_hor), 4
dev = torch.device("cuda", 0)
solver = BatchSolver(cfg, order="as_given", tail_promotion=0)
dt = DeviceTracker(cfg, B, device=0, solver=solver)
y0, pos_h, vel_h, static = scene_one(B, n_dyn, 5)
setup(dt, B, y0, static)
pos, vel = torch.from_numpy(pos_h).to(dev), torch.from_numpy(vel_h).to(dev)
k = torch.arange(1, N + 1, dtype=torch.float64, device=dev)
pred = torch.zeros(B, n_dyn, N, 6, dtype=torch.float64, device=dev)
pred[..., 2] = DYN_OBS_SIZE; pred[..., 3] = DYN_OBS_SIZE; pred[..., 5] = 1.0
rec = {k_: [] for k_ in ("evals", "status", "outer", "inner", "ms")}
guess = None
for t in range(ticks):
    pred[..., 0] = pos[..., None, 0] + vel[..., None, 0] * k
    pred[..., 1] = pos[..., None, 1] + vel[..., None, 1] * k
    dt.set_dynamic_constraints(pred)
    out = dt.step(initial_guess=guess)
    torch.cuda.synchronize()
    rec["evals"].append(solver.last_eval_counts(B, stream=torch.cuda.current_stream().cuda_stream)[0].copy())
    rec["status"].append(out["status"].cpu().numpy().copy()); rec["outer"].append(out["outer_it"].cpu().numpy().copy())
    rec["inner"].append(out["inner_it"].cpu().numpy().copy()); rec["ms"].append(solver.last_timing()["solve_ms"])
    if warm:
        u = out["u"].view(B, N, 2)
        guess = torch.cat([u[:, 1:], u[:, -1:]], dim=1).reshape(B, 2 * N).contiguous()
    pos += vel
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
f = os.path.join(ROOT, "gpurun_out", f"hint_probe_{'warm' if warm else 'cold'}.npz")
np.savez_compressed(f, **{k_: np.array(v) for k_, v in rec.items()})
print("wrote", f, "solve ms per tick:", " ".join(f"{m:.0f}" for m in rec["ms"]))
