"""Closed loop of the DQN-boosted MPC for B robots (reference: src/main.py decision modes 0 / 1 / 2) on scene 1 with the
'medium' unexpected box and a crossing dynamic obstacle.   usage: python tools/hybrid_loop.py [B] [max_steps]"""
import importlib, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from trajtrack_mpcndqn_rlboost_amd import MpcConfig
from trajtrack_mpcndqn_rlboost_amd.dqn import QNetwork
hybrid = importlib.import_module("trajtrack_mpcndqn_rlboost_amd.hybrid")
device_hybrid = importlib.import_module("trajtrack_mpcndqn_rlboost_amd.device_hybrid")
metrics = importlib.import_module("trajtrack_mpcndqn_rlboost_amd.metrics")


def scene(rng, dynamic=True):
    y0 = 3.5 + rng.uniform(-0.3, 0.3)
    return dict(boundary=[(0.0, 0.0), (16.0, 0.0), (16.0, 10.0), (0.0, 10.0)],
                static=[[(0.0, 1.5), (0.0, 1.6), (9.0, 1.6), (9.0, 1.5)], [(0.0, 8.4), (0.0, 8.5), (9.0, 8.5), (9.0, 8.4)],
                        [(11.0, 1.5), (11.0, 1.6), (16.0, 1.6), (16.0, 1.5)], [(11.0, 8.4), (11.0, 8.5), (16.0, 8.5), (16.0, 8.4)],
                        [(7.2, 2.8), (7.2, 4.2), (8.8, 4.2), (8.8, 2.8)]],
                dynamic=[dict(p1=(10.0, 1.0), p2=(10.0, 9.0), freq=0.2, rx=0.8, ry=0.8, angle=0.0, corners=20)] if dynamic else [],
                start=[0.6, y0, 0.0, 0.0, 0.0], goal=[15.4, 3.5],
                path=[(0.6, y0), (15.4, 3.5)])


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    # BASELINE.json config 4: under torchrun every rank drives B robots on its own GPU (no collective on the data path;
    # the success counts are summed once at the end).  HYBRID_BACKEND=gloo lets ranks share one GPU for testing.
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    backend = os.environ.get("HYBRID_BACKEND", "nccl")
    dev = local if backend == "nccl" else local % torch.cuda.device_count()
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, rank=rank, world_size=world, **({"device_id": torch.device("cuda", dev)} if backend == "nccl" else {}))
    w = np.load(os.path.join(ROOT, "tests", "golden", "dqn_ray.npz"))
    q = QNetwork().load_arrays({k: w[k] for k in w.files if k.startswith("w")})
    cfg = MpcConfig(os.path.join(ROOT, "config", "mpc_longiter.yaml"))
    for mode, name in ((1, "pure MPC"), (2, "hybrid"), (0, "pure DQN")):
        rng = np.random.default_rng(3 + 1000 * rank)
        # modes 1 and 2: the whole tick on the device (HYBRID_HOST_TICK=1 selects the host loop for comparison)
        on_device = mode != 0 and not os.environ.get("HYBRID_HOST_TICK")
        cls = device_hybrid.DeviceHybrid if on_device else hybrid.BatchedHybrid
        run = cls(cfg, [scene(rng) for _ in range(B)], q, decision_mode=mode, device=dev)
        t0 = time.perf_counter()
        out = run.run(steps, record=True)
        dt = time.perf_counter() - t0
        goal_dist = np.hypot(out["states"][:, 0] - 15.4, out["states"][:, 1] - 3.5)
        if world > 1:
            tot = torch.tensor([float(out["success"].sum()), float(out["collided"].sum()), float(B)], dtype=torch.float64,
                               device=f"cuda:{dev}" if backend == "nccl" else "cpu")
            dist.all_reduce(tot)
            if rank == 0:
                print(f"{name:9s} fleet of {int(tot[2])} robots on {world} ranks: success {tot[0] / tot[2]:.3f}, collided {tot[1] / tot[2]:.3f}", flush=True)
        if rank != 0:
            continue
        print(f"{name:9s} B={B}: ticks {run.t}, {1e3 * dt / run.t:.1f} ms/tick, success {out['success'].mean():.2f}, "
              f"collided {out['collided'].mean():.2f}, mean steps {out['steps'].mean():.0f}, "
              f"mean goal distance {goal_dist.mean():.2f}, ticks tracking the DQN proposal {out['switch_ticks'].mean():.1f}", flush=True)
        m = metrics.Metrics({0: "dqn", 1: "mpc", 2: "hyb"}[mode])
        m.add_batch(out["record"], [s["static"] for s in run.scenes])
        print("          metrics (main_pre.py Metrics):", m.get_average(), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
