#!/usr/bin/env python3
"""DQN training on the batched HIP environment (BASELINE.json config 5 in miniature, one GPU per process).

    python tools/train_dqn.py [--gpus N] [--envs 4096] [--timesteps 4000000] [--gradient-steps 16] [--batch-size 256]
                              [--save DIR] [--resume DIR/checkpoint.pt] [--checkpoint-every STEPS]

`--save DIR` writes what the reference's run leaves behind (src/test_block_rl.py:73-76,89-96: EvalCallback's best_model and
model.save's final_model) -- best_model.pt whenever the mean return of a progress window improves, final_model.pt at the end --
plus checkpoint.pt (network, target network, Adam moments, counters, generator, replay buffer, environment state) every
`--checkpoint-every` environment steps; `--resume` continues such a checkpoint exactly where it stopped (rank 0's files; in a
multi-rank run every rank restores its own `checkpoint.rank<r>.pt`).

`--gpus N` without a launcher starts the N ranks itself (fresh child processes of torch.distributed.run, before anything in
this process touches a GPU); it exits non-zero when fewer than N devices are visible.  Rank 0 ends with ONE JSON line
(environment steps/s and Q-network updates/s of the whole job, ranks, backend; `double_q` says whether they are double-Q updates).

Every environment is scene 1 of the reference (src/pkg_dqn/utils/map.py:292-305) with the 'medium' box and a periodic
obstacle, start pose jittered per environment; the reference path is the straight line start -> goal (an input; the
reference gets it from A*).  Under torch.distributed (torchrun, backend nccl = RCCL) every rank trains on its own
environments and the gradients are summed by one flat all-reduce per update.  Prints throughput and the return curve."""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rl_env = importlib.import_module("trajtrack_mpcndqn_rlboost_amd.rl_env")
dqn_train = importlib.import_module("trajtrack_mpcndqn_rlboost_amd.dqn_train")


def scene(rng):
    y0 = 3.5 + rng.uniform(-1.0, 1.0)
    th = rng.uniform(-0.5, 0.5)
    return rl_env.make_map(
        boundary=[(0.0, 0.0), (16.0, 0.0), (16.0, 10.0), (0.0, 10.0)],
        static=[[(0.0, 1.5), (0.0, 1.6), (9.0, 1.6), (9.0, 1.5)], [(0.0, 8.4), (0.0, 8.5), (9.0, 8.5), (9.0, 8.4)],
                [(11.0, 1.5), (11.0, 1.6), (16.0, 1.6), (16.0, 1.5)], [(11.0, 8.4), (11.0, 8.5), (16.0, 8.5), (16.0, 8.4)],
                [(7.2, 2.8), (7.2, 4.2), (8.8, 4.2), (8.8, 2.8)]],
        dynamic=[dict(p1=(10.0, 1.0), p2=(10.0, 9.0), freq=0.2, rx=0.8, ry=0.8, angle=0.0, corners=20)],
        start=[0.6, y0, th, 0.0, 0.0], goal=[15.4, 3.5], path=[(0.6, y0), (6.4, 5.4), (9.6, 5.4), (15.4, 3.5)])


def self_launch(gpus: int) -> int:
    if torch.cuda.device_count() < gpus:
        print(f"train_dqn.py: --gpus {gpus} but only {torch.cuda.device_count()} GPU(s) visible", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--timesteps", type=int, default=4_000_000)
    ap.add_argument("--gradient-steps", type=int, default=16)
    ap.add_argument("--batch-size", type=int, default=256)
    ap.add_argument("--double-q", action="store_true")
    ap.add_argument("--graph", action="store_true", help="replay the update from captured HIP graphs (multi-rank: two graphs around the all-reduce)")
    ap.add_argument("--save", default=None, help="directory for best_model.pt / final_model.pt / checkpoint.pt")
    ap.add_argument("--resume", default=None, help="checkpoint.pt of an earlier run with the same arguments")
    ap.add_argument("--checkpoint-every", type=int, default=0, help="environment steps between checkpoints (0: only at the end)")
    args = ap.parse_args()
    if os.environ.get("WORLD_SIZE") is None and args.gpus > 1:
        sys.exit(self_launch(args.gpus))
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus and (world > 1 or args.gpus > 1):
        sys.exit(f"train_dqn.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    rng = np.random.default_rng(100 + rank)
    torch.manual_seed(0)
    env = rl_env.BatchedRaysEnv([scene(rng) for _ in range(args.envs)], device=local, max_episode_steps=400)
    trainer = dqn_train.DqnTrainer(device=f"cuda:{local}", double_q=args.double_q, lr=1e-3)
    learner = dqn_train.DqnLearner(env, trainer, buffer_size=2_000_000, learning_starts=4 * args.envs,
                                   batch_size=args.batch_size, train_freq=4, gradient_steps=args.gradient_steps,
                                   target_update_interval=200_000, exploration_fraction=0.3, use_graph=args.graph,
                                   track_episodes=False)
    ck_name = "checkpoint.pt" if rank == 0 else f"checkpoint.rank{rank}.pt"
    if args.resume:
        learner.load(args.resume if rank == 0 else os.path.join(os.path.dirname(args.resume), ck_name))
        if rank == 0:
            print(f"  resumed at {learner.num_timesteps} environment steps, {learner.trainer.num_updates} updates", flush=True)
    if args.save:
        os.makedirs(args.save, exist_ok=True)
    marks, t0 = [], time.perf_counter()
    best = [-float("inf")]

    def progress(lr):
        if lr.num_timesteps // (args.timesteps // 10) > len(marks):
            n, r, ok = float(lr.ep_count), float(lr.ep_return_sum), float(lr.ep_success_sum)      # one sync per mark
            pn, pr, pok = marks[-1][4:] if marks else (0.0, 0.0, 0.0)
            dn = max(n - pn, 1.0)
            marks.append((lr.num_timesteps, (r - pr) / dn, (ok - pok) / dn, time.perf_counter() - t0, n, r, ok))
            if rank == 0:
                print(f"  {marks[-1][0]:>9d} steps  mean return {marks[-1][1]:8.2f}  success {marks[-1][2]:.2f}  {marks[-1][3]:6.1f} s", flush=True)
                if args.save and marks[-1][1] > best[0]:       # EvalCallback(best_model_save_path=...)
                    best[0] = marks[-1][1]
                    learner.save_model(os.path.join(args.save, "best_model.pt"))

    if args.save and args.checkpoint_every > 0:
        stats = None
        while learner.num_timesteps < args.timesteps:
            stats = learner.learn(args.timesteps, callback=progress, stop_at=learner.num_timesteps + args.checkpoint_every)
            learner.save(os.path.join(args.save, ck_name))
    else:
        stats = learner.learn(args.timesteps, callback=progress)
    if args.save:
        learner.save(os.path.join(args.save, ck_name))
        if rank == 0:
            learner.save_model(os.path.join(args.save, "final_model.pt"))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rank == 0:
        print(f"envs/rank {args.envs} x ranks {world}: {world * stats['timesteps'] / dt:.3e} environment steps/s incl. "
              f"{stats['updates']} updates of batch {args.batch_size} ({stats['updates'] / dt:.0f} updates/s), "
              f"episodes {stats['episodes']}, final mean return {stats['mean_return']:.2f}, success rate {stats['success_rate']:.2f}")
        print(json.dumps({"metric": "DQN online training: environment steps/s (batched HIP environment + "
                                    + ("double-Q" if args.double_q else "DQN") + " updates)",
                          "value": world * stats["timesteps"] / dt, "unit": "environment steps/s", "n_gpus": world,
                          "rccl_ranks": dist.get_world_size() if world > 1 else 1, "backend": "nccl" if world > 1 else None,
                          "updates_per_s": stats["updates"] / dt, "gradient_all_reduce_floats": 1177 if world > 1 else 0,
                          "update_path": "two HIP graphs around one flat RCCL all-reduce" if (args.graph and world > 1) else
                                         ("one HIP graph" if args.graph else "eager"),
                          "config": {"workload": "BASELINE.json config 5 (scene 1, medium box, periodic obstacle)",
                                     "envs_per_gpu": args.envs, "timesteps_per_gpu": int(stats["timesteps"]),
                                     "batch_size": args.batch_size, "gradient_steps": args.gradient_steps, "double_q": bool(args.double_q)},
                          "success_rate": stats["success_rate"], "mean_return": stats["mean_return"]}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
