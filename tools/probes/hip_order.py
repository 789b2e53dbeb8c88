"""Probe: a process that loads libmpcgpu.so before importing torch must still end up with ONE HIP runtime."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from trajtrack_mpcndqn_rlboost_amd import BatchSolver, MpcConfig, scenes
cfg = MpcConfig()
bs = BatchSolver(cfg, device=0)
sc = scenes.make_batch(cfg, 4, n_dyn=2, seed=7)
print("solve ok", bs.solve(sc["p"]).status)
print([l.split()[-1] for l in open("/proc/self/maps") if "amdhip" in l or "hsa-runtime" in l][::8])
import torch
print("torch sees GPU:", torch.cuda.is_available(), torch.cuda.device_count())
x = torch.ones(4, device="cuda"); print("tensor ok", float(x.sum()))
print(sorted({l.split()[-1] for l in open("/proc/self/maps") if "amdhip" in l or "hsa-runtime" in l}))
