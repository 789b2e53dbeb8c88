#!/usr/bin/env python3
"""Benchmark of the hot path: batched receding-horizon NMPC solves on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one batch: B independent cold-start solves (u0 = 0, what every
reference call site does: src/interface_mpc.py:82 passes initial_guess=None) of the metric configuration
named in BASELINE.json -- mpc_default.yaml, N_hor = 20, 8 dynamic obstacles -- on B = 131072 robots per GPU
(`--batch`; BASELINE.json prescribes no batch, SURVEY.md 8(d) words the metric at 8192 per GPU), with the
parameter vectors already resident in HBM.  A solve takes 0.1-0.2 s of device time and 4096 run concurrently; the
last problems of a launch finish on a draining GPU, which costs about 0.1 s per launch whatever the batch.  The
headline `value` is the plain leg: K launches of the whole shard, one after the other on one stream, each timed by
HIP events (that duration feeds `roofline`).  `config.batch_sweep` repeats the same workload at the reference
batches 32768 (round 1's bench batch) and 8192 (the metric batch), the latter also PIPELINED: two half shards on
two handles / two streams, every launch enqueued without a host synchronisation (mpcgpu_reserve_shape), so that
the next launch fills the GPU while the previous one drains -- the tail is then paid once per K steps.

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
                   stream IN THIS RUN; `flops` is the executed f64 flop rate from the in-kernel evaluation counters
                   of this run.  What needs hardware counters (`traffic`, `secondary`) is measured in this run as well
                   when rocprofv3 is there (N = 1): after the timed legs, three `rocprofv3 --pmc` passes (SQ counters,
                   FETCH_SIZE, WRITE_SIZE: one pass each, never combined with a trace) over a child bench.py on the
                   same parameter vectors, rebuilt by tools/roofline.py -- `measured_in_run: true`.  Without the
                   profiler (or with --no-pmc) the fields come from profiles/r06_roofline_bench.json -- the committed
                   passes on this very workload -- and are marked `measured_in_run: false` with the reason.
  config.convergent -- the same step on the "passing" scene family (same N, obstacle counts and batch; a
                   collision-free plan exists), where about half of the solves converge: the headline family is the
                   one SURVEY.md 8(d) prescribes and it is cap-limited (see status_histogram).
  config.avoidance -- the same step on the "avoidance" family (scenes.FAMILIES: 1-3 discs COVER the reference path, the box covers
                   it in 30 % of the problems, the detour leads to the free side of the corridor): the problems this MPC is for.
  config.closed_loop -- DeviceTracker at 8192 robots on scene 1 with 4 constant-velocity discs (src/main.py:160-222 semantics): 30
                   ticks after 5 warm-up ticks, cold and warm start, MPCGPU_OPT_ORDER off and on -- the hints are REAL there
                   (tick k orders tick k + 1); tools/closed_loop.py.
  config.batch_sweep -- see above; every batch also `ordered_perfect_hints`.
  config.ordered_perfect_hints -- the headline batch with the library's default dispatch order (MPCGPU_OPT_ORDER = 1: longest
                   first by the evaluation counts of the previous call).  A bench step repeats the SAME batch, which makes those
                   hints perfect: an UPPER BOUND for a receding-horizon loop (config.closed_loop shows what real hints give).
                   The headline `value` itself starts the problems as given.
  config.closed_loop.realtime_robots_per_gpu -- the largest fleet (multiple of 512 robots) whose WORST tick of that run stays within
                   the sampling time ts = 0.2 s of the yaml, cold and warm start (`realtime`: the sizes tried, the per-tick times).
  config.metric_batch -- the 8192-robot batch SURVEY.md 8(d) words the metric at, plain launches as given (copied from batch_sweep).
  converged_solves_per_s, converged_fraction (top level, next to `value`) -- what `value` is made of: the headline family is the one
                   SURVEY.md 8(d) prescribes and nearly all of its solves end at the iteration caps; these two say so without a look
                   into `config` (the same figures of the convergent families are in config.convergent / config.avoidance).
  config.penalty_stall -- the reading of the ALM penalty-stall rule this run solved with (yaml key solver_penalty_stall; DESIGN.md
                   section 3), and `config.other_stall_reading`: two launches of the headline batch under the other reading.
  config.host_boundary -- `mpcgpu_solve_batch` (HOST pointers -- the call the reference's plugin makes, trajectory_generator.py:318) at
                   the metric batch 8192, from pageable and from page-locked memory, and one robot per call: solves/s and the share of
                   the call that is not the solve kernels (copies + synchronisation).
  psi_evals_per_s -- psi evaluations per second of the headline leg: the workload-independent rate (families with other iteration
                   counts compare on it).
  The side legs (convergent, avoidance, ordered, batch_sweep, closed_loop, counter passes, cpu_baseline) run with ONE rank only
  (world == 1) unless --full is given: an N-rank run is the headline leg and nothing else.
  cpu_baseline  -- the oracle (plain-C restatement, "port") on the host cores this process may use, on a bounded
                   sample.  Its answers double as the checker of THIS run: `cpu_baseline.parity_on_sample` compares the control
                   sequences the timed launches wrote for the same problems (converged pairs: max |du| against the 1e-3 tolerance;
                   agreement on which problems converge; the cap-limited rest is reported, not judged); `parity_on_families` does the same,
                   untimed, for the first 1024 problems of the convergent side legs, where half of the solves converge.  The headline
                   family alone yields a handful of converged pairs, so `parity_on_sample.converged_pairs_total` adds the families' pairs
                   (>= 256) and `feeds` says which sample contributes how many.
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
ROOFLINE_JSON = os.path.join(ROOT, "profiles", "r06_roofline_bench.json")
SWEEP_BATCHES = (32768, 8192)   # reference batches reported next to the headline (round 1's bench batch; SURVEY.md 8(d)'s metric batch)
PARITY_SAMPLE = 1024  # problems per convergent side-leg family handed to the oracle in the cpu_baseline leg (seconds of CPU)


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
    ap.add_argument("--no-sweep", action="store_true", help="skip the reference-batch legs (profiling runs)")
    ap.add_argument("--no-pmc", action="store_true", help="do not re-measure the counter-based roofline fields in this run")
    ap.add_argument("--no-closed-loop", action="store_true", help="skip the closed-loop leg")
    ap.add_argument("--no-capacity", action="store_true", help="skip the real-time capacity search of the closed-loop leg")
    ap.add_argument("--no-host-boundary", action="store_true", help="skip the host-pointer leg (mpcgpu_solve_batch)")
    ap.add_argument("--penalty-stall", choices=("either", "both"), default=None,
                    help="reading of the ALM penalty-stall rule (default: the yaml / library default)")
    ap.add_argument("--side-batch", type=int, default=32768, help="problems per GPU of the convergent / avoidance legs")
    ap.add_argument("--full", action="store_true", help="run the side legs on every rank of a multi-GPU run as well")
    ap.add_argument("--p-file", default=None, help=argparse.SUPPRESS)   # counter passes: the parent's parameter vectors (.npy)
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


PMC_PASSES = (("pmc_sq", ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_INSTS_SALU", "SQ_WAVES", "GRBM_GUI_ACTIVE")),
              ("pmc_fetch", ("FETCH_SIZE",)),
              ("pmc_write", ("WRITE_SIZE",)))


def pmc_in_run(args, p_host):
    """The counter-based roofline fields, measured IN THIS RUN: three `rocprofv3 --pmc` passes (SQ counters, FETCH_SIZE and
    WRITE_SIZE each in a pass of its own -- never combined with a trace) over a CHILD bench.py on the very same parameter vectors
    (one warm-up + one timed launch of the whole shard, no side legs), then tools/roofline.py's rebuild on the CSVs.  The
    profiled program is `python3 bench.py ...` itself, placed directly after `--`.  Returns (derived dict, None) or
    (None, reason)."""
    import shutil
    import tempfile
    rp = shutil.which("rocprofv3")
    if rp is None:
        return None, "rocprofv3 not on PATH"
    raw = tempfile.mkdtemp(prefix="bench_pmc_", dir="/tmp")
    pfile = os.path.join(raw, "p.npy")
    try:
        np.save(pfile, p_host)
        with open(os.path.join(raw, "workload.json"), "w") as fh:
            json.dump({"N_hor": args.horizon, "n_dyn": args.n_dyn, "batch_per_gpu": args.batch, "steps": 1, "warmup": 1,
                       "command": "bench.py's own counter passes (pmc_in_run)"}, fh)
        child = [sys.executable, os.path.abspath(__file__), "--steps", "1", "--warmup", "1", "--cpu-seconds", "0",
                 "--no-convergent", "--no-sweep", "--no-pmc", "--no-closed-loop", "--batch", str(args.batch), "--n-dyn", str(args.n_dyn),
                 "--horizon", str(args.horizon), "--p-file", pfile] + (["--penalty-stall", args.penalty_stall] if args.penalty_stall else [])
        env = dict(os.environ, TMPDIR="/tmp")
        env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
        t0 = time.perf_counter()
        for name, counters in PMC_PASSES:
            cmd = [rp, "--pmc", *counters, "--output-format", "csv", "-d", os.path.join(raw, name), "-o", name, "--"] + child
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=240)
            if r.returncode != 0:
                return None, f"{name}: rocprofv3 exit code {r.returncode}: " + r.stdout.decode(errors="replace")[-200:]
            found = None
            for d, _, files in os.walk(os.path.join(raw, name)):
                for f in files:
                    if f.endswith("counter_collection.csv"):
                        found = os.path.join(d, f)
            if found is None:
                return None, f"{name}: no counter_collection.csv"
            shutil.copy(found, os.path.join(raw, name + "_counter_collection.csv"))
        from tools.roofline import rebuild
        d = rebuild(raw, os.path.join(raw, "roofline.json"))
        d["pmc_passes_wall_s"] = time.perf_counter() - t0
        return d, None
    except Exception as e:   # noqa: BLE001 -- the bench line must come out whatever the profiler does
        return None, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(raw, ignore_errors=True)


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

    def reserve_shape(self, **kw):
        pass

    def reserve_batch(self, B):
        pass

    def set_order(self, order):
        pass


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

    cfg_kw = {} if args.penalty_stall is None else {"solver_penalty_stall": args.penalty_stall}
    cfg = MpcConfig(N_hor=args.horizon, **cfg_kw)
    N, B = cfg.N_hor, args.batch

    def new_solver(order="as_given", config=None):
        """`order`: MPCGPU_OPT_ORDER.  Every leg that feeds `value` starts the problems in the order given; the library's default
        (longest first by the previous call's evaluation counts) is measured in legs of its own (`ordered`): a bench step repeats
        the SAME batch, so those hints are perfect -- an upper bound of what a receding-horizon loop gets from its last tick."""
        if stub:
            return StubSolver(config or cfg)
        from trajtrack_mpcndqn_rlboost_amd import BatchSolver
        return BatchSolver(config or cfg, device=dev_index, order=order)

    def new_out(b):
        return dict(u=torch.empty(b, 2 * N, dtype=torch.float64, device=dev),
                    cost=torch.empty(b, dtype=torch.float64, device=dev),
                    status=torch.empty(b, dtype=torch.int32, device=dev),
                    inner_it=torch.empty(b, dtype=torch.int32, device=dev),
                    outer_it=torch.empty(b, dtype=torch.int32, device=dev),
                    f2norm=torch.empty(b, dtype=torch.float64, device=dev))
    solver = new_solver()
    out = new_out(B)
    stream = None if stub else torch.cuda.current_stream().cuda_stream

    def barrier():
        if world > 1:
            dist.barrier()
        if not stub:
            torch.cuda.synchronize()

    def reserve_like_last(sv, b):
        """After one automatic call (count read-back): promise exactly the shape it found, so that the timed launches read
        nothing back -- the same kernel instantiation, no host synchronisation between the compaction and the solve."""
        sh = sv.last_shape()
        sv.reserve_shape(max_static=sh["max_static"], max_fleet=sh["max_fleet"], max_dyn=sh["max_dyn"],
                         var_shape=not sh.get("shape_const", False), axis_aligned=sh.get("axis_aligned", False))
        sv.reserve_batch(b)

    def timed_leg(p, steps, warmup, sv=None, o=None):
        """W untimed + K timed passes over the batch `p`, one launch after the other; returns the per-rank elapsed time (max
        over ranks is taken by the caller), the mean kernel times and the per-problem outcome of the last pass."""
        sv = sv or solver
        o = o or out
        b = int(p.shape[0])
        sv.release_shape() if hasattr(sv, "release_shape") else None
        sv.solve_device(p, o, stream=stream)          # automatic rule: finds the batch's shape
        reserve_like_last(sv, b)
        for _ in range(max(warmup - 1, 0)):
            sv.solve_device(p, o, stream=stream)
        barrier()
        k_ms, p_ms, t_ms = [], [], []
        t0 = time.perf_counter()
        for _ in range(steps):
            sv.solve_device(p, o, stream=stream)
            t = sv.last_timing()          # HIP events recorded on `stream` around the two kernels
            k_ms.append(t.get("main_ms", t["solve_ms"])); p_ms.append(t["prep_ms"]); t_ms.append(t.get("tail_ms", 0.0))
        barrier()
        mine = time.perf_counter() - t0
        n_psi, n_grad = sv.last_eval_counts(b, stream)
        promo = sv.last_tail_promotion(stream) if hasattr(sv, "last_tail_promotion") else (0, 0)
        return dict(elapsed=mine, kernel_ms=float(np.mean(k_ms)), prep_ms=float(np.mean(p_ms)), tail_ms=float(np.mean(t_ms)), tail_promotion=promo,
                    status=o["status"].cpu().numpy().copy(), inner=o["inner_it"].cpu().numpy().copy(),
                    n_psi=n_psi, n_grad=n_grad)

    def pipelined_leg(p, steps, warmup):
        """The same K passes with the batch split into two half shards on two handles and two streams; nothing is read back
        and nothing waits on the host inside the timed region, so launch k + 1 of one stream fills the compute units that
        launch k of the other stream is draining.  (The library sees the other handle's launch in flight and keeps the continuation of
        the tail promotion behind its own throughput kernel: MPCGPU_OPT_TAIL_CONCURRENT applies to one launch at a time.)"""
        b = int(p.shape[0])
        halves = [p[: b // 2], p[b // 2:]]
        svs = [new_solver(), new_solver()]
        outs = [new_out(int(h.shape[0])) for h in halves]
        if stub:
            streams = [None, None]
        else:
            ts = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
            streams = [t.cuda_stream for t in ts]
        for sv, h, o, st in zip(svs, halves, outs, streams):
            sv.solve_device(h, o, stream=st)
            if not stub:
                torch.cuda.synchronize()
            reserve_like_last(sv, int(h.shape[0]))
        for _ in range(max(warmup - 1, 0)):
            for sv, h, o, st in zip(svs, halves, outs, streams):
                sv.solve_device(h, o, stream=st)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            for sv, h, o, st in zip(svs, halves, outs, streams):
                sv.solve_device(h, o, stream=st)
        barrier()
        mine = time.perf_counter() - t0
        status = np.concatenate([o["status"].cpu().numpy() for o in outs])
        for sv in svs:
            sv.close() if hasattr(sv, "close") else None
        return dict(elapsed=mine, status=status)

    def over_ranks(x: float):
        """(max over ranks, list of every rank's value)."""
        if world == 1:
            return x, [x]
        t = torch.tensor([x], dtype=torch.float64, device=red_dev)
        allv = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allv, t)
        vals = [float(v.item()) for v in allv]
        return max(vals), vals

    # Scene generation.  ONE host buffer serves every family: the first touch of a fresh 2.8 GB numpy array (page faults) costs
    # more than generating its content, so the buffer is allocated once -- page-locked when there is a GPU, which also populates
    # it in bulk -- and every family is generated into (a slice of) it and uploaded.
    t_gen = time.perf_counter()
    host_buf = None
    if not args.p_file:
        try:
            host_buf = torch.empty((B, cfg.num_params), dtype=torch.float64, pin_memory=not stub).numpy()
        except RuntimeError:     # no page-locked memory to be had (cgroup limit): pageable, page faults and all
            host_buf = np.empty((B, cfg.num_params))

    def family_on_device(family, b, seed):
        scenes.make_family(cfg, b, family, n_dyn=args.n_dyn, seed=seed, out=host_buf[:b])
        return torch.from_numpy(host_buf[:b]).to(dev, copy=True)
    # every rank owns its own B robots of the global scene set (weak scaling): global problem g = rank*B + i
    if args.p_file:   # counter pass of a parent bench.py: its parameter vectors, not a second scene generation
        sc = {"p": np.load(args.p_file)}
        assert sc["p"].shape == (B, cfg.num_params), sc["p"].shape
        p = torch.from_numpy(sc["p"]).to(dev)
    else:
        p = family_on_device("benchmark", B, 1234 + 7919 * rank)
        sc = {"p": None}            # the host copy is reused by the other families; cpu_baseline / counter passes read p back
    gen_s = time.perf_counter() - t_gen
    side = world == 1 or args.full   # side legs: one rank (or --full)
    leg = timed_leg(p, args.steps, args.warmup)
    # the control sequences of the first problems of the headline batch, for the parity figures next to the CPU baseline (below)
    head_u = None if stub else out["u"][:16384].cpu().numpy().copy()
    elapsed, per_rank_s = over_ranks(leg["elapsed"])

    def ordered_leg(pb, steps, sv=None, o=None):
        """The same launches with the problems started longest first (hints = the evaluation counts of the previous step)."""
        own = sv is None
        sv = sv or new_solver("longest_first")
        o = o or new_out(int(pb.shape[0]))
        ol = timed_leg(pb, steps, 2, sv, o)      # the second warm-up launch already runs with hints
        o_el, _ = over_ranks(ol["elapsed"])
        b = int(pb.shape[0])
        item = {"value": world * b * steps / o_el, "unit": "solves/s", "steps": steps, "ms_per_step": 1e3 * o_el / steps,
                "kernel_ms": ol["kernel_ms"] + ol["tail_ms"], "status_histogram": np.bincount(ol["status"], minlength=3).tolist(),
                "how": "MPCGPU_OPT_ORDER = 1 (the library's default): problems started longest first by the evaluation counts of the "
                       "previous step; the step repeats the same batch, so the hints are PERFECT: an upper bound (config.closed_loop "
                       "has the figure with the real hints of a receding-horizon loop)"}
        if own and hasattr(sv, "close"):
            sv.close()
        return item

    ordered_head = None
    if side and not args.no_sweep:
        solver.set_order("longest_first")
        ordered_head = ordered_leg(p, min(args.steps, 5), solver, out)
        solver.set_order("as_given")

    # side legs on other scene families (same N_hor, obstacle counts; `--side-batch` problems per GPU)
    Bs = min(B, args.side_batch)
    side_legs = {}
    if side and not args.no_convergent and not args.p_file:
        for fam, seed in (("passing", 4321), ("avoidance", 8642)):
            pc = family_on_device(fam, Bs, seed + 7919 * rank)
            sv_s, o_s = new_solver(), new_out(Bs)
            cleg = timed_leg(pc, min(args.steps, 5), min(args.warmup, 1), sv_s, o_s)
            c_elapsed, _ = over_ranks(cleg["elapsed"])
            side_legs[fam] = dict(leg=cleg, elapsed=c_elapsed, steps=min(args.steps, 5))
            if not stub:   # the first problems of this family with what the timed launches wrote for them: checked in the cpu_baseline leg
                side_legs[fam]["sample"] = dict(p=pc[:PARITY_SAMPLE].cpu().numpy().copy(), u=o_s["u"][:PARITY_SAMPLE].cpu().numpy().copy(),
                                                status=cleg["status"][:PARITY_SAMPLE].copy())
            if hasattr(sv_s, "close"):
                sv_s.close()
            del pc, o_s
    conv = side_legs.get("passing")

    # reference batches of the same workload (the first b problems of this rank's shard): plain launches, and -- for the
    # metric batch -- the pipelined form.  Bounded step counts: these legs must not dominate the run.
    sweep = []
    if side and not args.no_sweep:
        side_steps, side_warm = min(args.steps, 5), min(args.warmup, 1)
        for b in SWEEP_BATCHES:
            if b >= B:
                continue
            pb = p[:b].contiguous()
            sv, ob = new_solver(), new_out(b)
            sl = timed_leg(pb, side_steps, side_warm, sv, ob)
            s_el, _ = over_ranks(sl["elapsed"])
            item = {"batch_per_gpu": b, "steps": side_steps,
                    "plain": {"value": world * b * side_steps / s_el, "unit": "solves/s", "ms_per_step": 1e3 * s_el / side_steps,
                              "kernel_ms": sl["kernel_ms"] + sl["tail_ms"], "status_histogram": np.bincount(sl["status"], minlength=3).tolist(),
                              "psi_evals_per_s": world * float(sl["n_psi"].sum()) * side_steps / s_el,
                              "tail_promotion": {"capacity": sl["tail_promotion"][0], "promoted_last_step": sl["tail_promotion"][1]}}}
            if hasattr(sv, "close"):
                sv.close()
            item["ordered_perfect_hints"] = ordered_leg(pb, side_steps)
            if b == SWEEP_BATCHES[-1]:
                pl = pipelined_leg(pb, 4 * side_steps, side_warm)
                p_el, _ = over_ranks(pl["elapsed"])
                item["pipelined"] = {"value": world * b * 4 * side_steps / p_el, "unit": "solves/s", "steps": 4 * side_steps,
                                     "ms_per_step": 1e3 * p_el / (4 * side_steps),
                                     "status_histogram": np.bincount(pl["status"], minlength=3).tolist(),
                                     "how": "two half shards, two handles, two streams; launches enqueued back to back "
                                            "(reserved shape: no read-back, no host synchronisation inside the timed region)"}
            sweep.append(item)

    # the headline batch under the OTHER reading of the penalty-stall rule: two launches, same problems
    other_reading = None
    if side and not args.no_sweep and not args.p_file:
        other = "both" if getattr(cfg, "solver_penalty_stall", "either") == "either" else "either"
        cfg_o = MpcConfig(N_hor=args.horizon, solver_penalty_stall=other)
        sv_o = new_solver(config=cfg_o)
        ol = timed_leg(p, 2, 1, sv_o, out)
        o_el, _ = over_ranks(ol["elapsed"])
        other_reading = {"penalty_stall": other, "value": world * B * 2 / o_el, "unit": "solves/s", "steps": 2, "ms_per_step": 1e3 * o_el / 2,
                         "kernel_ms": ol["kernel_ms"] + ol["tail_ms"], "status_histogram": np.bincount(ol["status"], minlength=3).tolist(),
                         "converged_fraction": float((ol["status"] == 0).mean()), "mean_inner_iterations": float(ol["inner"].mean()),
                         "mean_psi_evaluations": float(ol["n_psi"].mean()),
                         "psi_evals_per_s": world * float(ol["n_psi"].sum()) * 2 / o_el}
        if hasattr(sv_o, "close"):
            sv_o.close()

    # The boundary the reference's plugin actually crosses: HOST pointers (solver.run(p) with Python lists, trajectory_generator.py:272-275,318)
    # -> mpcgpu_solve_batch: copy in, compaction + solve, copy out, synchronous.  The metric batch from pageable and from page-locked
    # memory, and one robot per call (the reference's own pattern).  `not_the_solve_kernels` = 1 - (compaction + solve kernel time by
    # HIP events) / wall time of the call: copies, launches and the synchronisation.
    host_leg = None
    if side and not args.no_host_boundary and not stub and not args.p_file:
        hb = min(8192, B)
        sv_h = new_solver()
        p_page = p[:hb].cpu().numpy().copy()
        pin_t = torch.empty((hb, cfg.num_params), dtype=torch.float64, pin_memory=True)
        pin_t.copy_(p[:hb]); torch.cuda.synchronize()

        def host_out(n, pinned):
            mk = (lambda shape, dt: torch.empty(shape, dtype=dt, pin_memory=True).numpy()) if pinned else \
                 (lambda shape, dt: np.empty(shape, dtype={torch.float64: np.float64, torch.int32: np.int32}[dt]))
            return dict(u=mk((n, 2 * N), torch.float64), cost=mk((n,), torch.float64), status=mk((n,), torch.int32),
                        inner_it=mk((n,), torch.int32), outer_it=mk((n,), torch.int32))
        host_leg = {"what": "mpcgpu_solve_batch (HOST pointers; copies in, compaction, solve, copies out, synchronous) on the first problems of the "
                            "headline batch; bytes over PCIe per solve: " + str(8 * cfg.num_params + 8 * 2 * N + 8 + 4 + 4 + 4),
                    "batch": hb}
        for kind, pp in (("pageable", p_page), ("pinned", pin_t.numpy())):
            oo = host_out(hb, kind == "pinned")
            sv_h.solve_into(pp, oo)                                   # warm-up: buffers, shape
            walls, kern = [], []
            for _ in range(3):
                t0 = time.perf_counter()
                sv_h.solve_into(pp, oo)
                walls.append(time.perf_counter() - t0)
                t = sv_h.last_timing()
                kern.append(1e-3 * (t["prep_ms"] + t["solve_ms"]))
            w, k = float(np.mean(walls)), float(np.mean(kern))
            host_leg[kind] = {"value": hb / w, "unit": "solves/s", "ms_per_call": 1e3 * w, "kernel_ms": 1e3 * k,
                              "not_the_solve_kernels": 1.0 - k / w, "status_histogram": np.bincount(oo["status"], minlength=3).tolist()}
        one = host_out(1, False)
        n_one = 48
        walls, kern = [], []
        sv_h.solve_into(p_page[:1], one)
        for i in range(n_one):
            t0 = time.perf_counter()
            sv_h.solve_into(p_page[i:i + 1], one)
            walls.append(time.perf_counter() - t0)
            t = sv_h.last_timing()
            kern.append(1e-3 * (t["prep_ms"] + t["solve_ms"]))
        host_leg["one_robot_per_call"] = {"calls": n_one, "value": n_one / float(np.sum(walls)), "unit": "solves/s",
                                          "ms_per_call_median": 1e3 * float(np.median(walls)), "ms_per_call_max": 1e3 * float(np.max(walls)),
                                          "kernel_ms_median": 1e3 * float(np.median(kern)),
                                          "not_the_solve_kernels": 1.0 - float(np.sum(kern)) / float(np.sum(walls)),
                                          "what": "the reference's call pattern (src/interface_mpc.py:82-88: one solver.run per robot and tick): "
                                                  "pageable host vectors, the latency kernel (four wavefronts per problem)"}
        sv_h.close()
        del pin_t

    closed = None
    if side and not args.no_closed_loop and not stub and not args.p_file and N == 20:
        from tools.closed_loop import device_closed_loop
        closed = {"workload": "DeviceTracker, 8192 robots per GPU, scene 1 (corridor + inflated box on the path, src/pkg_dqn/utils/map.py:292-305) "
                              "with 4 constant-velocity discs (src/main.py:77-85), one solve per control tick (src/main.py:160-222), 30 timed "
                              "ticks after 5 warm-up ticks; wall time per tick by HIP events around the whole tick (predictions, window, "
                              "assembly, solve, rollouts); nothing read back inside the loop",
                  "runs": []}
        for warm in (False, True):
            for order in ("as_given", "longest_first"):
                r = device_closed_loop(cfg, 8192, 30, 5, 4, warm, order, device=dev_index)
                r.pop("_final_states")
                per_tick = r.pop("status_histogram_per_tick")
                r["status_histogram_first_tick"], r["status_histogram_last_tick"] = per_tick[0], per_tick[-1]
                closed["runs"].append(r)
        if not args.no_capacity:
            # the reference prints its solve time per control step (src/main.py:230-238); for a fleet the question is how many robots
            # ONE GPU serves with every tick inside the sampling time ts of the yaml (0.2 s)
            from tools.closed_loop import realtime_capacity
            cap = {("warm" if warm else "cold"): realtime_capacity(cfg, warm=warm, device=dev_index) for warm in (False, True)}
            closed["realtime_robots_per_gpu"] = {k: v["robots"] for k, v in cap.items()}
            closed["realtime"] = cap
        barrier()

    if rank == 0:
        status, inner = leg["status"], leg["inner"]
        n_psi, n_grad = leg["n_psi"], leg["n_grad"]
        # the whole solve of the batch on the launch stream: the throughput kernel AND what is left of the continuation behind it (the
        # numerators below -- bytes, flops -- count every problem of the batch, the promoted ones included)
        k_ms = leg["kernel_ms"] + leg["tail_ms"]
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
            # workload-independent rate: psi (+ grad psi) evaluations per second of this rank's shard x ranks (families with other
            # iteration counts compare on this, not on solves/s)
            "psi_evals_per_s": world * float(n_psi.sum()) * args.steps / elapsed,
            # what `value` is made of (this scene family is cap-limited: SURVEY.md 8(d) prescribes it; config.convergent / avoidance have the others)
            "converged_solves_per_s": world * n_conv * args.steps / elapsed,
            "converged_fraction": n_conv / B,
            "config": {"workload": f"mpc_default.yaml N_hor={N}, {args.n_dyn} dynamic obstacles (r=1.6 m discs crossing "
                                   "the path, SURVEY.md 8(d)), 5 static boxes, cold start u0=0, "
                                   f"batch={B} robots per GPU; K plain launches on one stream, problems started in the order given "
                                   "(config.ordered: the library's default order; batch_sweep: the reference batches 32768 and "
                                   "8192, plain / ordered / pipelined)",
                       "batch_per_gpu": B, "N_hor": N, "n_dyn": args.n_dyn, "parallelism": f"shard{world}",
                       "penalty_stall": getattr(cfg, "solver_penalty_stall", "either"),
                       "mean_inner_iterations": float(inner.mean()),
                       "mean_psi_evaluations": float(n_psi.mean()), "mean_grad_evaluations": float(n_grad.mean()),
                       "status_histogram": np.bincount(status, minlength=3).tolist(),
                       "converged_fraction": n_conv / B,
                       "converged_solves_per_s": world * n_conv * args.steps / elapsed,
                       "lds_bytes_per_wavefront": solver.last_shape()["lds_bytes"],
                       "wavefronts_per_simd": solver.last_shape()["waves_per_simd"],
                       "tail_promotion": {"what": "MPCGPU_OPT_TAIL_PROMOTION (library default): the last problems of a plain launch "
                                                  "move to the latency kernel at the start of their next inner problem, which runs them on a side stream "
                                                  "while the launch drains (MPCGPU_OPT_TAIL_CONCURRENT); bitwise the same results",
                                          "capacity": leg["tail_promotion"][0], "promoted_last_step": leg["tail_promotion"][1],
                                          "continuation_ms_after_the_throughput_kernel": leg["tail_ms"]}},
        }
        if conv is not None:
            cl = conv["leg"]
            line["config"]["convergent"] = {
                "workload": f"same N_hor and obstacle counts, batch {Bs} per GPU; 'passing' family (scenes.FAMILIES: dyn_clearance=0.1, "
                            "box_clearance=0.3): discs and box beside the path, a collision-free plan exists",
                "batch_per_gpu": Bs,
                "value": world * Bs * conv["steps"] / conv["elapsed"], "unit": "solves/s", "steps": conv["steps"],
                "ms_per_step": 1e3 * conv["elapsed"] / conv["steps"], "kernel_ms": cl["kernel_ms"] + cl["tail_ms"],
                "status_histogram": np.bincount(cl["status"], minlength=3).tolist(),
                "converged_fraction": float((cl["status"] == 0).mean()),
                "mean_inner_iterations": float(cl["inner"].mean()),
                "mean_psi_evaluations": float(cl["n_psi"].mean()),
                "converged_solves_per_s": world * int((cl["status"] == 0).sum()) * conv["steps"] / conv["elapsed"]}
        if "avoidance" in side_legs:
            av = side_legs["avoidance"]
            al = av["leg"]
            line["config"]["avoidance"] = {
                "workload": f"same N_hor and obstacle counts, batch {Bs} per GPU; 'avoidance' family (scenes.FAMILIES): 1-3 of the 8 discs "
                            "(hard radius 1.6 m) cover the reference path by 0.1-0.6 m, the inflated box covers it by 2-30 cm in 30 % of "
                            "the problems, the detour leads to the free side of the corridor; previous speed 0.8-1.2 m/s",
                "batch_per_gpu": Bs,
                "value": world * Bs * av["steps"] / av["elapsed"], "unit": "solves/s", "steps": av["steps"],
                "ms_per_step": 1e3 * av["elapsed"] / av["steps"], "kernel_ms": al["kernel_ms"] + al["tail_ms"],
                "status_histogram": np.bincount(al["status"], minlength=3).tolist(),
                "converged_fraction": float((al["status"] == 0).mean()),
                "mean_inner_iterations": float(al["inner"].mean()),
                "mean_psi_evaluations": float(al["n_psi"].mean()),
                "converged_solves_per_s": world * int((al["status"] == 0).sum()) * av["steps"] / av["elapsed"]}
        if other_reading is not None:
            line["config"]["other_stall_reading"] = other_reading
        if host_leg is not None:
            line["config"]["host_boundary"] = host_leg
        if closed is not None:
            line["config"]["closed_loop"] = closed
        if sweep:
            line["config"]["batch_sweep"] = sweep
            mb = [it for it in sweep if it["batch_per_gpu"] == 8192]
            if mb:      # SURVEY.md 8(d) words the metric at 8192 robots per GPU: that figure, plain launches as given, at the top level of config
                line["config"]["metric_batch"] = {"batch_per_gpu": 8192, "what": "plain launches, problems started as given (batch_sweep has the other forms)",
                                                  **mb[0]["plain"]}
        if ordered_head is not None:
            line["config"]["ordered_perfect_hints"] = ordered_head
        line["config"]["scene_generation_s"] = gen_s
        if not stub:
            from tools.roofline import flops_per_solve_kernel_launch, load_pmc_for
            pmc = load_pmc_for(ROOFLINE_JSON, N, args.n_dyn, B)
            p_host = sc["p"] if sc["p"] is not None else p.cpu().numpy()
            pmc_here, pmc_err = (None, "disabled (--no-pmc)") if (args.no_pmc or world > 1) else pmc_in_run(args, p_host)
            in_run = pmc_here is not None
            if in_run:
                pmc = pmc_here
            shape = solver.last_shape()
            flops = flops_per_solve_kernel_launch(N, shape["max_static"], shape["max_fleet"], shape["max_dyn"], n_psi, n_grad)
            tf = flops / (k_ms * 1e-3) / 1e12
            roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": pmc["traffic_bytes_per_launch"] if pmc else None,
                    "kernel": "solve_kernel_pair", "kernel_ms": k_ms, "throughput_kernel_ms": leg["kernel_ms"], "prep_kernel_ms": leg["prep_ms"],
                    "tail_kernel": "solve_kernel_team (continuation of the tail promotion: the last problems of the launch; it starts on a side stream while "
                                   "solve_kernel_pair drains -- tail_kernel_ms is the part of it AFTER that kernel has ended, by HIP events; kernel_ms = throughput_kernel_ms + tail_kernel_ms is what achieved / frac / flops divide by)",
                    "tail_kernel_ms": leg["tail_ms"],
                    "algorithmic_bytes_per_solve": algo_bytes // B,
                    "measured_in_run": {"achieved": True, "kernel_ms": True, "frac": True, "flops": True,
                                        "traffic": in_run, "wasted_traffic_ratio": in_run, "traffic_GBps": in_run,
                                        "secondary": in_run},
                    "wasted_traffic_ratio": (pmc["traffic_bytes_per_launch"] / algo_bytes) if pmc else None,
                    "traffic_GBps": (pmc["traffic_bytes_per_launch"] / (pmc["kernel_avg_ms_kernel_trace"] * 1e-3) / 1e9) if pmc else None,
                    "traffic_source": (("this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, one pass each, over a child "
                                        f"bench.py on the same parameter vectors ({pmc_here['pmc_passes_wall_s']:.0f} s for the three "
                                        "counter passes)") if in_run else
                                       (os.path.relpath(ROOFLINE_JSON, ROOT) + ": separate rocprofv3 --pmc passes on this "
                                        f"workload; NOT re-measured in this run ({pmc_err})") if pmc else
                                       f"not measured: {pmc_err}; no committed counter passes for this workload"),
                    "note": "state is register/LDS/L2 resident: the kernel is VALU-issue bound (see secondary); traffic = "
                            "L2<->fabric bytes of the kernel's own cold state (L-BFGS ring, previous iterate), DESIGN.md section 2",
                    "flops": {"bound": "valu_f64_flops", "achieved": tf, "peak": FP64_VECTOR_PEAK_TF, "unit": "TFLOP/s",
                              "frac": tf / FP64_VECTOR_PEAK_TF, "measured_in_run": True,
                              "source": "static f64-flop table per evaluation (tools/roofline.py: by horizon and active rows) "
                                        "x the in-kernel psi / grad evaluation counters of this run / this run's kernel time"},
                    "secondary": None}
            if pmc:
                # VALU issue: instructions per launch from the PMC pass of this workload over THAT pass's kernel time, against
                # one wave64 instruction per 4 cycles per SIMD
                roof["secondary"] = {"bound": "valu_issue", "measured_in_run": in_run,
                                     "frac": pmc["valu_issue_frac_of_peak"],
                                     "valu_busy_frac_pmc": pmc["valu_busy_fraction"],
                                     "resident_waves_per_simd": pmc["resident_waves_per_simd"],
                                     "valu_instructions_per_solve": pmc["valu_instructions_per_launch"] / B,
                                     "kernel_ms_of_the_pmc_pass": pmc["kernel_avg_ms_kernel_trace"],
                                     "source": "this run: rocprofv3 --pmc SQ_* pass over a child bench.py" if in_run
                                               else os.path.relpath(ROOFLINE_JSON, ROOT)}
            line["roofline"] = roof
        if args.cpu_seconds > 0 and world == 1 and not stub:
            line["cpu_baseline"] = cpu_baseline(cfg, p[:16384].cpu().numpy(), args.cpu_seconds, gpu_u=head_u, gpu_status=leg["status"][:16384],
                                                family_samples={f: v["sample"] for f, v in side_legs.items() if "sample" in v})
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def host_cores():
    """What this process may use: scheduler affinity, cgroup CPU quota, and the SMT layout of the machine."""
    affinity = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:                      # cgroup v2: "<quota> <period>" or "max <period>"
            q, per = fh.read().split()
            if q != "max":
                quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
                q, per = float(fq.read()), float(fp.read())
                if q > 0:
                    quota = q / per
        except (OSError, ValueError):
            pass
    threads_per_core = None
    try:
        sib = cores = None
        with open("/proc/cpuinfo") as fh:
            for ln in fh:
                if ln.startswith("siblings") and sib is None:
                    sib = int(ln.split(":")[1])
                elif ln.startswith("cpu cores") and cores is None:
                    cores = int(ln.split(":")[1])
        if sib and cores:
            threads_per_core = sib // cores
    except (OSError, ValueError):
        pass
    usable = affinity if quota is None else max(1, min(affinity, int(quota + 0.5)))
    return dict(logical_cpus=os.cpu_count() or 1, affinity=affinity, cgroup_quota_cpus=quota,
                threads_per_core=threads_per_core, usable=usable)


def cpu_baseline(cfg, p_all, budget_s, gpu_u=None, gpu_status=None, family_samples=None):
    """The oracle (C restatement of the same algorithm, kind = "port") on the host cores this process may use, on the
    first S problems of the very same workload; S is calibrated so the run takes about `budget_s` seconds.  With the GPU's control
    sequences and statuses of the same problems (`gpu_u`, `gpu_status`: the timed launches' own outputs) the oracle's answers -- here
    the checker -- give the parity figures of THIS run: `parity_on_sample`; `family_samples` (the first problems of the convergent
    side legs with the GPU's answers) get the same check, untimed: `parity_on_families` -- there half of the solves converge."""
    import oracle
    ocfg = oracle.OracleConfig.from_dict(cfg.solver_dict())
    hc = host_cores()
    cores = hc["usable"]
    probe = min(len(p_all), max(cores, 32))
    t = time.perf_counter()
    oracle.solve_batch(ocfg, p_all[:probe], nthreads=cores)
    dt = time.perf_counter() - t
    S = int(min(len(p_all), max(probe, probe * 0.8 * budget_s / max(dt, 1e-3))))
    S = max(cores, (S // cores) * cores)
    S = min(S, len(p_all))
    t = time.perf_counter()
    u_cpu, _, res, used = oracle.solve_batch(ocfg, p_all[:S], nthreads=cores)
    dt = time.perf_counter() - t
    smt = hc["threads_per_core"]
    def compare(u_c, st_c, u_g, st_g):
        n = len(st_c)
        st_c, st_g = np.asarray(st_c)[:n], np.asarray(st_g)[:n]
        du = np.max(np.abs(np.asarray(u_c)[:n] - np.asarray(u_g)[:n]), axis=1)
        both = (st_c == 0) & (st_g == 0)
        far = both & (du > 1e-3)      # converged on both sides, yet apart: two local minima (a detour on either side of an obstacle;
        near = both & ~far            # the oracle against its own 1-ulp twin shows the same: profiles/r06_fuzz_parity.txt)
        return {"problems": int(n), "converged_on_both_sides": int(both.sum()),
                "max_abs_du_on_them": float(du[both].max()) if both.any() else None, "tolerance": 1e-3,
                "pairs_beyond_tolerance": int(far.sum()), "max_abs_du_of_the_pairs_within": float(du[near].max()) if near.any() else None,
                "same_converged_or_not": float(np.mean((st_c == 0) == (st_g == 0))),
                "median_abs_du_of_the_cap_limited": float(np.median(du[~both])) if (~both).any() else None}
    parity = None
    if gpu_u is not None and gpu_status is not None and len(gpu_u) >= S:
        parity = compare(u_cpu[:S], np.asarray(res["status"])[:S], gpu_u, gpu_status)
        parity["note"] = ("GPU control sequences of the timed launches against the oracle's on the same problems; solves that run "
                          "into the iteration cap are chaotic in any float64 implementation (DESIGN.md section 3): compared on the "
                          "converged pairs, reported for the others")
    families = {}
    for fam, smp in (family_samples or {}).items():
        u_f, _, res_f, _ = oracle.solve_batch(ocfg, smp["p"], nthreads=cores)
        families[fam] = compare(u_f, res_f["status"], smp["u"], smp["status"])
    if parity is not None:
        # The headline family is cap-limited: a handful of converged pairs.  The tolerance is carried by the convergent families at the same
        # horizon and obstacle counts (the first problems of the timed side legs): their pairs are added here, and `feeds` says who gave what.
        feeds = {"headline_sample": parity["converged_on_both_sides"], **{f: v["converged_on_both_sides"] for f, v in families.items()}}
        allp = [parity] + list(families.values())
        within = [v["max_abs_du_of_the_pairs_within"] for v in allp if v["max_abs_du_of_the_pairs_within"] is not None]
        parity["converged_pairs_total"] = int(sum(feeds.values()))
        parity["feeds"] = feeds
        parity["pairs_beyond_tolerance_total"] = int(sum(v["pairs_beyond_tolerance"] for v in allp))
        parity["max_abs_du_of_the_pairs_within_total"] = max(within) if within else None
    return {"value": S / dt, "unit": "solves/s", "cores": used, "kind": "port", "parity_on_sample": parity, "parity_on_families": families or None,
            "per_core_solves_per_s": S / dt / used,
            "host": {"logical_cpus": hc["logical_cpus"], "sched_affinity": hc["affinity"],
                     "cgroup_quota_cpus": hc["cgroup_quota_cpus"], "threads_per_core": smt,
                     "note": (f"{used} OpenMP threads = the logical CPUs this process may use; "
                              + (f"SMT-{smt}: {used // smt if smt else used} physical cores" if smt and smt > 1 else
                                 "no SMT reported" if smt == 1 else "SMT layout unknown"))},
            "sample": f"first {S} problems of the same batch, OpenMP over problems, {dt:.1f} s wall, "
                      f"mean inner iterations {float(res['inner_iters'].mean()):.0f}"}


if __name__ == "__main__":
    main()
