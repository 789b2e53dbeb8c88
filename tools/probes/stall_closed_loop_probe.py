"""Probe (GPU box): closed-loop behaviour of the scanner scenes and the hybrid loop under the two penalty-stall readings."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from trajtrack_mpcndqn_rlboost_amd import MpcConfig
replay = importlib.import_module("scanner_replay")
for stall in ("either", "both"):
    for s in (1, 2, 3, 4, 5):
        o = replay.replay(s, 64, cfg=MpcConfig(solver_penalty_stall=stall), jitter=0.05)
        print(f"[{stall}] scene {s}: arrived {o['arrived'].mean():.3f} ticks {o['ticks']} min_core min {o['min_core'].min():.3f} median {np.median(o['min_core']):.3f} "
              f"min_pair min {o['min_pair'].min():.3f} median {np.median(o['min_pair']):.3f} statuses {o['status_histogram'].tolist()}", flush=True)
hybrid = importlib.import_module("trajtrack_mpcndqn_rlboost_amd.hybrid")
loop = importlib.import_module("hybrid_loop")
from trajtrack_mpcndqn_rlboost_amd.dqn import QNetwork
w = np.load(os.path.join(ROOT, "tests", "golden", "dqn_ray.npz"))
q = QNetwork().load_arrays({k: w[k] for k in w.files if k.startswith("w")})
for stall in ("either", "both"):
    cfg = MpcConfig(os.path.join(ROOT, "config", "mpc_longiter.yaml"), solver_penalty_stall=stall)
    for B in (8, 32):
        rng = np.random.default_rng(3)
        scenes = [loop.scene(rng) for _ in range(B)]
        for mode in (1, 2):
            out = hybrid.BatchedHybrid(cfg, scenes, q, decision_mode=mode).run(200)
            print(f"[{stall}] hybrid B {B} mode {mode}: done {out['done'].mean():.3f} collided {out['collided'].mean():.3f} success {out['success'].mean():.3f} "
                  f"switch ticks >0 {(out['switch_ticks'] > 0).mean():.3f}", flush=True)
