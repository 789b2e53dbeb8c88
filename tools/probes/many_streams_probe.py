import sys, os
sys.path.insert(0, os.getcwd())
import torch
from trajtrack_mpcndqn_rlboost_amd import MpcConfig, BatchSolver
from tools.closed_loop import device_closed_loop
cfg = MpcConfig("config/mpc_default.yaml")
extra = int(sys.argv[1]) if len(sys.argv) > 1 else 6
keep = [BatchSolver(cfg) for _ in range(extra)]
ts = [torch.cuda.Stream() for _ in range(extra)]
for t in ts:
    with torch.cuda.stream(t):
        torch.zeros(8, device="cuda").sum()
torch.cuda.synchronize()
for order in ("as_given", "longest_first"):
    r = device_closed_loop(cfg, 8192, 30, 5, 4, False, order)
    print(f"extra handles/streams {extra}: order {order}: {r['ms_per_tick']:.1f} ms/tick, worst {r['ms_per_tick_min_max'][1]:.1f}", flush=True)
