#!/usr/bin/env python3
"""Benchmark of the hot path: batched receding-horizon NMPC solves on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one batch: B independent cold-start solves (u0 = 0, what every
reference call site does: src/interface_mpc.py:82 passes initial_guess=None) of the metric configuration
named in BASELINE.json -- mpc_default.yaml, N_hor = 20, 8 dynamic obstacles -- on B = 131072 robots per GPU,
with the parameter vectors already resident in HBM.  (A solve takes 0.1-0.2 s of device time and 4096 run
concurrently; the last problems of a launch finish on a draining GPU, which costs about 0.12 s per launch whatever
the batch: measured 24.3 / 27.2 / 28.3 / 29.1 thousand solves/s at B = 16384 / 32768 / 65536 / 131072
(profiles/r02_batch_scaling.txt; round 1 benchmarked B = 32768: 25.2 thousand then, 27.2 thousand now).  `--batch` selects
other sizes.)

Multi-GPU (N > 1): one process per GPU; the batch shards across ranks with no data-path collective (weak
scaling: every rank owns its own B robots); RCCL is used only for the barrier, the max-over-ranks time and
the gather of the per-rank rates.  Launched either by the driver (`python -m torch.distributed.run
--nproc-per-node N ... bench.py --gpus N`: RANK / LOCAL_RANK / WORLD_SIZE in the environment) or DIRECTLY as
`python bench.py --gpus N`: with WORLD_SIZE unset this process starts the N ranks itself through
torch.distributed.run as a CHILD process -- before anything here has touched the GPU -- and exits with the
child's code.  It exits non-zero when fewer than N devices are present, and when WORLD_SIZE disagrees with
--gpus.

Rank 0 prints ONE JSON line (contract in the task description) with extra objects:
  roofline      -- dominant kernel (solve_kernel) against the HBM roofline, timed with HIP events on the launch
                   stream; `secondary` is the bound that actually binds (VALU issue), `flops` the executed f64
                   flop rate from the in-kernel evaluation counters; PMC-derived fields come from
                   profiles/r02_roofline_bench.json and are used ONLY when that file was collected on this very
                   workload (tools/roofline.py rebuilds it from the raw rocprofv3 CSVs).
  config.convergent -- the same step on the "passing" scene family (same N, obstacle counts and batch; a
                   collision-free plan exists), where most solves converge: the headline family is the one
                   SURVEY.md 8(d) prescribes and it is cap-limited (see status_histogram).
  cpu_baseline  -- the oracle (plain-C restatement, "port") on the host cores, on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VECTOR_PEAK_TF = 78.6   # MI355X vector FP64: 256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz
ROOFLINE_JSON = os.path.join(ROOT, "profiles", "r02_roofline_bench.json")


def shard(total: int, rank: int, world: int):
    """Contiguous chunk [lo, hi) of `total` items owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=131072, help="problems per GPU per step")
    ap.add_argument("--n-dyn", type=int, default=8)
    ap.add_argument("--horizon", type=int, default=20)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU-baseline budget (0 disables)")
    ap.add_argument("--no-convergent", action="store_true", help="skip the second leg (profiling runs)")
    return ap.parse_args(argv)


def self_launch(args) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as a child process group.  Nothing in this
    process has initialised the GPU yet (torch.cuda.device_count() does not), and the ranks are fresh interpreters --
    no exec of a process that holds a GPU context."""
    backend = os.environ.get("MPCGPU_BENCH_BACKEND", "nccl")
    if backend == "nccl" and not os.environ.get("MPCGPU_BENCH_STUB"):
        import torch
        n_dev = torch.cuda.device_count()
        if n_dev < args.gpus:
            print(f"bench.py: --gpus {args.gpus} but only {n_dev} GPU(s) visible", file=sys.stderr)
            return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


class StubSolver:
    """MPCGPU_BENCH_STUB=1 (tests of the launch / reduction plumbing on machines without a GPU): same interface as
    BatchSolver.solve_device, no solve.  The JSON line then says data = "stub" and carries no roofline."""

    def __init__(self, cfg):
        self.n = 2 * int(cfg.N_hor)

    def solve_device(self, p, out, stream=None):
        out["u"].zero_(); out["cost"].zero_(); out["status"].fill_(1)
        out["inner_it"].fill_(7); out["outer_it"].fill_(1)
        time.sleep(0.01)

    def last_timing(self):
        return dict(prep_ms=0.0, solve_ms=10.0)

    def last_eval_counts(self, B, stream=None):
        return np.full(B, 3, np.int32), np.full(B, 2, np.int32)

    def last_shape(self):
        return dict(max_static=0, max_fleet=0, max_dyn=0, lds_bytes=0, waves_per_simd=0)


def main():
    args = parse_args()
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        sys.exit(self_launch(args))
    world = int(world_env or "1")
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (launch with --nproc-per-node {args.gpus}, "
                         "or run `python bench.py --gpus N` without a launcher)")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import torch
    from trajtrack_mpcndqn_rlboost_amd import MpcConfig, scenes

    stub = bool(os.environ.get("MPCGPU_BENCH_STUB"))
    # MPCGPU_BENCH_BACKEND=gloo lets ranks share one GPU (or none, with the stub) to exercise the N > 1 path (testing)
    backend = os.environ.get("MPCGPU_BENCH_BACKEND", "nccl")
    if stub:
        dev_index, dev = 0, torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
        n_dev = torch.cuda.device_count()
        if backend == "nccl" and n_dev < world:
            raise SystemExit(f"bench.py: {world} ranks but only {n_dev} GPU(s) visible")
        dev_index = local_rank if backend == "nccl" else local_rank % n_dev
        torch.cuda.set_device(dev_index)
        dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    red_dev = dev if (backend == "nccl" and not stub) else torch.device("cpu")

    cfg = MpcConfig(N_hor=args.horizon)
    N, B = cfg.N_hor, args.batch
    if stub:
        solver = StubSolver(cfg)
    else:
        from trajtrack_mpcndqn_rlboost_amd import BatchSolver
        solver = BatchSolver(cfg, device=dev_index)
    out = dict(u=torch.empty(B, 2 * N, dtype=torch.float64, device=dev),
               cost=torch.empty(B, dtype=torch.float64, device=dev),
               status=torch.empty(B, dtype=torch.int32, device=dev),
               inner_it=torch.empty(B, dtype=torch.int32, device=dev),
               outer_it=torch.empty(B, dtype=torch.int32, device=dev),
               f2norm=torch.empty(B, dtype=torch.float64, device=dev))
    stream = None if stub else torch.cuda.current_stream().cuda_stream

    def barrier():
        if world > 1:
            dist.barrier()
        if not stub:
            torch.cuda.synchronize()

    def timed_leg(p, steps, warmup):
        """W untimed + K timed passes over the batch `p`; returns the per-rank elapsed time (max over ranks is taken by
        the caller), the mean kernel times and the per-problem outcome of the last pass."""
        for _ in range(warmup):
            solver.solve_device(p, out, stream=stream)
        barrier()
        k_ms, p_ms = [], []
        t0 = time.perf_counter()
        for _ in range(steps):
            solver.solve_device(p, out, stream=stream)
            t = solver.last_timing()          # HIP events recorded on `stream` around the two kernels
            k_ms.append(t["solve_ms"]); p_ms.append(t["prep_ms"])
        barrier()
        mine = time.perf_counter() - t0
        n_psi, n_grad = solver.last_eval_counts(B, stream)
        return dict(elapsed=mine, kernel_ms=float(np.mean(k_ms)), prep_ms=float(np.mean(p_ms)),
                    status=out["status"].cpu().numpy().copy(), inner=out["inner_it"].cpu().numpy().copy(),
                    n_psi=n_psi, n_grad=n_grad)

    def over_ranks(x: float):
        """(max over ranks, list of every rank's value)."""
        if world == 1:
            return x, [x]
        t = torch.tensor([x], dtype=torch.float64, device=red_dev)
        allv = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allv, t)
        vals = [float(v.item()) for v in allv]
        return max(vals), vals

    # every rank owns its own B robots of the global scene set (weak scaling): global problem g = rank*B + i
    sc = scenes.make_batch(cfg, B, n_dyn=args.n_dyn, seed=1234 + 7919 * rank)
    p = torch.from_numpy(sc["p"]).to(dev)
    leg = timed_leg(p, args.steps, args.warmup)
    elapsed, per_rank_s = over_ranks(leg["elapsed"])

    conv = None
    if not args.no_convergent:
        scc = scenes.make_batch(cfg, B, n_dyn=args.n_dyn, seed=4321 + 7919 * rank, dyn_clearance=0.1, box_clearance=0.3)
        pc = torch.from_numpy(scc["p"]).to(dev)
        cleg = timed_leg(pc, args.steps, args.warmup)
        c_elapsed, _ = over_ranks(cleg["elapsed"])
        conv = dict(leg=cleg, elapsed=c_elapsed)
        del pc

    if rank == 0:
        status, inner = leg["status"], leg["inner"]
        n_psi, n_grad = leg["n_psi"], leg["n_grad"]
        k_ms = leg["kernel_ms"]
        algo_bytes = (8 * cfg.num_params + 8 * 2 * N + 8 * 2 * N + 40) * B
        achieved = algo_bytes / (k_ms * 1e-3) / 1e9
        n_conv = int((status == 0).sum())
        line = {
            "metric": "MPC solves/sec (batch, N=20 horizon, 8 dyn obs)",
            "value": world * B * args.steps / elapsed,
            "unit": "solves/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "stub" if stub else "synthetic",
            "rccl_ranks": dist.get_world_size() if world > 1 else 1,
            "backend": backend if world > 1 else None,
            "per_rank_solves_per_s": [B * args.steps / t for t in per_rank_s],
            "config": {"workload": f"mpc_default.yaml N_hor={N}, {args.n_dyn} dynamic obstacles (r=1.6 m discs crossing "
                                   "the path, SURVEY.md 8(d)), 5 static boxes, cold start u0=0, "
                                   f"batch={B} robots per GPU (BASELINE.json metric configuration)",
                       "batch_per_gpu": B, "N_hor": N, "n_dyn": args.n_dyn, "parallelism": f"shard{world}",
                       "mean_inner_iterations": float(inner.mean()),
                       "mean_psi_evaluations": float(n_psi.mean()), "mean_grad_evaluations": float(n_grad.mean()),
                       "status_histogram": np.bincount(status, minlength=3).tolist(),
                       "converged_fraction": n_conv / B,
                       "converged_solves_per_s": world * n_conv * args.steps / elapsed,
                       "lds_bytes_per_wavefront": solver.last_shape()["lds_bytes"],
                       "wavefronts_per_simd": solver.last_shape()["waves_per_simd"]},
        }
        if conv is not None:
            cl = conv["leg"]
            line["config"]["convergent"] = {
                "workload": "same N_hor, obstacle counts and batch; 'passing' family (scenes.make_batch dyn_clearance=0.1, "
                            "box_clearance=0.3): discs and box beside the path, a collision-free plan exists",
                "value": world * B * args.steps / conv["elapsed"], "unit": "solves/s",
                "ms_per_step": 1e3 * conv["elapsed"] / args.steps, "kernel_ms": cl["kernel_ms"],
                "status_histogram": np.bincount(cl["status"], minlength=3).tolist(),
                "converged_fraction": float((cl["status"] == 0).mean()),
                "mean_inner_iterations": float(cl["inner"].mean()),
                "mean_psi_evaluations": float(cl["n_psi"].mean())}
        if not stub:
            from tools.roofline import flops_per_solve_kernel_launch, load_pmc_for
            pmc = load_pmc_for(ROOFLINE_JSON, N, args.n_dyn, B)
            shape = solver.last_shape()
            flops = flops_per_solve_kernel_launch(N, shape["max_static"], shape["max_fleet"], shape["max_dyn"], n_psi, n_grad)
            tf = flops / (k_ms * 1e-3) / 1e12
            roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": pmc["traffic_bytes_per_launch"] if pmc else None,
                    "kernel": "solve_kernel", "kernel_ms": k_ms, "prep_kernel_ms": leg["prep_ms"],
                    "algorithmic_bytes_per_solve": algo_bytes // B,
                    "wasted_traffic_ratio": (pmc["traffic_bytes_per_launch"] / algo_bytes) if pmc else None,
                    "note": "state is register/LDS/L2 resident: the kernel is VALU-issue bound (see secondary); traffic = "
                            "L2<->fabric bytes of the kernel's own cold state (L-BFGS ring, spill slots), DESIGN.md section 2",
                    "flops": {"bound": "valu_f64_flops", "achieved": tf, "peak": FP64_VECTOR_PEAK_TF, "unit": "TFLOP/s",
                              "frac": tf / FP64_VECTOR_PEAK_TF,
                              "source": "static f64-flop table per evaluation (tools/roofline.py: by horizon and active rows) "
                                        "x the in-kernel psi / grad evaluation counters of this run / this run's kernel time"},
                    "secondary": None}
            if pmc:
                # VALU issue: instructions per launch from the PMC pass of this workload / THIS run's kernel time, against
                # one wave64 instruction per 4 cycles per SIMD
                peak_ips = 256 * 4 * pmc["clock_GHz"] * 1e9 / 4.0
                ach_ips = pmc["valu_instructions_per_launch"] / (k_ms * 1e-3)
                roof["secondary"] = {"bound": "valu_issue", "achieved": ach_ips / 1e9, "peak": peak_ips / 1e9,
                                     "unit": "G wave-instructions/s", "frac": ach_ips / peak_ips,
                                     "valu_busy_frac_pmc": pmc["valu_busy_fraction"],
                                     "resident_waves_per_simd": pmc["resident_waves_per_simd"],
                                     "valu_instructions_per_solve": pmc["valu_instructions_per_launch"] / B,
                                     "source": os.path.relpath(ROOFLINE_JSON, ROOT)}
            line["roofline"] = roof
        if args.cpu_seconds > 0 and world == 1 and not stub:
            line["cpu_baseline"] = cpu_baseline(cfg, sc["p"], args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(cfg, p_all, budget_s):
    """The oracle (C restatement of the same algorithm, kind = "port") on all host cores, on the first S
    problems of the very same workload; S is calibrated so the run takes about `budget_s` seconds."""
    import oracle
    ocfg = oracle.OracleConfig.from_dict(cfg.solver_dict())
    cores = os.cpu_count() or 1
    probe = min(len(p_all), max(cores, 32))
    t = time.perf_counter()
    oracle.solve_batch(ocfg, p_all[:probe], nthreads=cores)
    dt = time.perf_counter() - t
    S = int(min(len(p_all), max(probe, probe * 0.8 * budget_s / max(dt, 1e-3))))
    S = max(cores, (S // cores) * cores)
    S = min(S, len(p_all))
    t = time.perf_counter()
    _, _, res, used = oracle.solve_batch(ocfg, p_all[:S], nthreads=cores)
    dt = time.perf_counter() - t
    return {"value": S / dt, "unit": "solves/s", "cores": used, "kind": "port",
            "sample": f"first {S} problems of the same batch, OpenMP over problems, {dt:.1f} s wall, "
                      f"mean inner iterations {float(res['inner_iters'].mean()):.0f}"}


if __name__ == "__main__":
    main()
