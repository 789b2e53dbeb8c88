#!/usr/bin/env python3
"""The concurrent continuation of the tail promotion (MPCGPU_OPT_TAIL_CONCURRENT, include/mpcgpu.h) against FOREIGN work on the device
(GPU box).  The library sees the launches of its own handles (another_launch_in_flight); it cannot see kernels the caller runs on other
streams (the DQN / hybrid tick's torch kernels) or another process on the same GPU.  Its waits are bounded and a sweep launch finishes what
a starved side stream leaves -- so foreign work may cost time, never results.  This tool measures exactly that:

  gemm     a torch stream kept busy with f32 GEMMs (a thread re-fills it) while `calls` solves of B problems run on another stream
  process  a second PROCESS (bench.py --steps 2 on 32768 problems) shares the GPU while the solves run
  quiet    the same solves alone (the baseline of the per-call times)

per mode and batch: every call's outputs against the bits of a solve without tail promotion, wall time per call (median, p90, max, max /
median), how many calls ran the continuation beside the launch, and the number of bounded waits that ended by their time limit
(mpcgpu_last_tail_timeouts).

usage: python tools/foreign_work_soak.py [--calls 200] [--batches 4096,8192] [--modes quiet,gemm,process] [--concurrent 1]"""
import argparse
import json
import os
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def gemm_load(stop, dev, n=4096, depth=6):
    """Keeps a torch stream `depth` GEMMs deep until `stop` is set."""
    import torch
    st = torch.cuda.Stream(device=dev)
    a = torch.randn(n, n, device=dev)
    b = torch.randn(n, n, device=dev)
    launched = 0
    with torch.cuda.stream(st):
        while not stop.is_set():
            for _ in range(depth):
                a = torch.mm(a, b).mul_(1.0 / n)       # keeps the values bounded
            launched += depth
            st.synchronize()
    return launched


def bench_child(batch=32768):
    """The second process: a short bench.py run on its own handle / context."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", str(batch), "--no-convergent", "--no-sweep",
           "--no-pmc", "--no-closed-loop", "--no-host-boundary", "--cpu-seconds", "0"]
    return subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT)


def soak(cfg, B, calls, mode, concurrent=True, family="passing", seed=4321, device=0, max_seconds=300.0):
    """Returns a dict of figures; raises AssertionError when a call's outputs differ from the reference bits."""
    import torch
    from trajtrack_mpcndqn_rlboost_amd import BatchSolver, scenes
    dev = torch.device("cuda", device)
    N = int(cfg.N_hor)
    p_h = scenes.make_family(cfg, B, family, n_dyn=8, seed=seed)["p"]
    ref = BatchSolver(cfg, device=device, tail_promotion=0, order="as_given")
    r0 = ref.solve(p_h)
    ref.close()
    want = dict(u=torch.from_numpy(r0.solution).to(dev), cost=torch.from_numpy(r0.cost).to(dev), status=torch.from_numpy(r0.status).to(dev),
                inner_it=torch.from_numpy(r0.num_inner_iterations).to(dev))
    p = torch.from_numpy(p_h).to(dev)
    out = dict(u=torch.empty(B, 2 * N, dtype=torch.float64, device=dev), cost=torch.empty(B, dtype=torch.float64, device=dev),
               status=torch.empty(B, dtype=torch.int32, device=dev), inner_it=torch.empty(B, dtype=torch.int32, device=dev),
               outer_it=torch.empty(B, dtype=torch.int32, device=dev))
    bs = BatchSolver(cfg, device=device)
    bs.set_tail_concurrent(bool(concurrent))
    solve_stream = torch.cuda.Stream(device=dev)
    st = solve_stream.cuda_stream
    bs.solve_device(p, out, stream=st); solve_stream.synchronize()
    sh = bs.last_shape()
    bs.reserve_shape(max_static=sh["max_static"], max_fleet=sh["max_fleet"], max_dyn=sh["max_dyn"], var_shape=not sh["shape_const"],
                     axis_aligned=sh["axis_aligned"])
    bs.reserve_batch(B)
    stop, worker, child, launched = threading.Event(), None, None, [0]
    if mode == "gemm":
        worker = threading.Thread(target=lambda: launched.__setitem__(0, gemm_load(stop, dev)), daemon=True)
        worker.start()
        time.sleep(0.2)
    elif mode == "process":
        child = bench_child()
    walls, beside, timeouts, moved = [], 0, 0, []
    t_begin = time.perf_counter()
    n_done = 0
    try:
        for i in range(calls):
            if mode == "process" and child.poll() is not None and i >= 8:
                break                                     # the other process is done: what follows would be a quiet run
            if time.perf_counter() - t_begin > max_seconds:
                break
            with torch.cuda.stream(solve_stream):          # (on the solve's own stream: a zeroing kernel on another stream could land after the solve)
                for t in out.values():
                    t.zero_()
            solve_stream.synchronize()
            t0 = time.perf_counter()
            bs.solve_device(p, out, stream=st)
            solve_stream.synchronize()
            walls.append(time.perf_counter() - t0)
            conc, to = bs.last_tail_timeouts(st)
            beside += int(conc); timeouts += to
            moved.append(bs.last_tail_promotion(st)[1])
            for k, w in want.items():
                assert torch.equal(out[k], w), f"call {i} ({mode}, B = {B}): {k} differs from the solve without tail promotion"
            n_done += 1
    finally:
        stop.set()
        if worker is not None:
            worker.join(timeout=60)
        child_rc, child_value = None, None
        if child is not None:
            try:
                so, se = child.communicate(timeout=600)
                child_rc = child.returncode
                for ln in so.splitlines():
                    if ln.startswith("{"):
                        child_value = json.loads(ln).get("value")
            except subprocess.TimeoutExpired:
                child.kill()
                child_rc = -9
        bs.close()
    w = np.array(walls)
    return dict(mode=mode, batch=B, calls=n_done, ms_median=1e3 * float(np.median(w)), ms_p90=1e3 * float(np.quantile(w, 0.9)), ms_max=1e3 * float(w.max()),
                max_over_median=float(w.max() / np.median(w)), beside_the_launch=beside, timeouts=timeouts, promoted_median=float(np.median(moved)),
                gemms_launched=launched[0], child_rc=child_rc, child_solves_per_s=child_value, bitwise="all calls")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--calls", type=int, default=200)
    ap.add_argument("--batches", default="4096,8192")
    ap.add_argument("--modes", default="quiet,gemm,process")
    ap.add_argument("--concurrent", type=int, default=1)
    args = ap.parse_args()
    from trajtrack_mpcndqn_rlboost_amd import MpcConfig
    cfg = MpcConfig(N_hor=20)
    print(__doc__.split("usage:")[0])
    print(f"N_hor = 20, 'passing' family (converging and cap-limited solves side by side), 8 discs; MPCGPU_OPT_TAIL_CONCURRENT = {args.concurrent}; "
          f"up to {args.calls} calls per line")
    for B in [int(x) for x in args.batches.split(",")]:
        for mode in args.modes.split(","):
            r = soak(cfg, B, args.calls, mode, bool(args.concurrent))
            print(f"  B {B:5d} {mode:8s}: {r['calls']:3d} calls bitwise equal; per call median {r['ms_median']:7.1f} ms, p90 {r['ms_p90']:7.1f}, max {r['ms_max']:7.1f} "
                  f"(max / median {r['max_over_median']:.2f}); continuation beside the launch in {r['beside_the_launch']} calls, promoted (median) "
                  f"{r['promoted_median']:.0f}; waits ended by their time limit: {r['timeouts']}"
                  + (f"; GEMMs on the other stream: {r['gemms_launched']}" if mode == "gemm" else "")
                  + (f"; other process: exit code {r['child_rc']}, its solves/s {r['child_solves_per_s']}" if mode == "process" else ""), flush=True)


if __name__ == "__main__":
    main()
