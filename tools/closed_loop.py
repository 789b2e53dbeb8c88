#!/usr/bin/env python3
"""Closed-loop run of the batched tracker on the MI355X: B robots drive the reference's scene 1 (corridor, inflated box on the
path, src/pkg_dqn/utils/map.py:292-305) while discs cross it; one GPU solve per control tick -- the loop of src/main.py:160-222
(pure-MPC branch) / src/scenario_simulator.py:211-250 for B robots at once.

`device_closed_loop` is the DEVICE-RESIDENT form (DeviceTracker: window search, assembly, solve and rollouts enqueued on one
stream, the constant-velocity predictions of src/main.py:77-85 formed by torch kernels, nothing read back inside the timed
ticks); bench.py reports it as `config.closed_loop`.  The hints of MPCGPU_OPT_ORDER are REAL here: the evaluation counts of tick
k order tick k + 1.

usage: closed_loop.py [B] [ticks] [n_dyn] [warm] [host | capacity]   ("host": the host-assembly loop of rounds 1-3, BatchedTracker;
"capacity": the largest fleet per GPU whose worst tick stays within the sampling time, realtime_capacity)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WALLS = [[(0.0, 1.5), (0.0, 1.6), (9.0, 1.6), (9.0, 1.5)], [(0.0, 8.4), (0.0, 8.5), (9.0, 8.5), (9.0, 8.4)],
         [(11.0, 1.5), (11.0, 1.6), (16.0, 1.6), (16.0, 1.5)], [(11.0, 8.4), (11.0, 8.5), (16.0, 8.5), (16.0, 8.4)]]
BOX = [(7.5, 3.0), (7.5, 4.0), (8.5, 4.0), (8.5, 3.0)]


def inflate(poly, m=0.8):
    xs, ys = [x for x, _ in poly], [y for _, y in poly]
    return [(min(xs) - m, min(ys) - m), (max(xs) + m, min(ys) - m), (max(xs) + m, max(ys) + m), (min(xs) - m, max(ys) + m)]


def scene_one(B, K, seed=5):
    """Start rows, the common path / goal, static polygons, disc positions and velocities (one private copy of the scene per robot)."""
    rng = np.random.default_rng(seed)
    y0 = rng.uniform(3.0, 4.0, B)
    pos = np.stack([rng.uniform(10.0, 14.0, (B, K)), rng.uniform(4.5, 7.0, (B, K))], axis=-1)
    vel = np.stack([rng.uniform(-0.12, -0.04, (B, K)), rng.uniform(-0.03, 0.03, (B, K))], axis=-1)
    static = [inflate(w) for w in WALLS] + [inflate(BOX)]
    return y0, pos, vel, static


def setup(tracker, B, y0, static):
    for i in range(B):
        tracker.initialization(i, np.array([0.6, y0[i], 0.0]), np.array([15.4, 3.5, 0.0]),
                               [(0.6, y0[i]), (6.0, 5.6), (10.0, 5.6), (15.4, 3.5)], "work")
        tracker.update_static_constraints(i, static)


def device_closed_loop(cfg, B=8192, ticks=30, warmup_ticks=5, n_dyn=4, warm=False, order="longest_first", device=0, seed=5,
                       tail_promotion=None):
    """Returns a dict: per-tick wall time (HIP events around the whole tick), solves/s, status histogram of every timed tick,
    mean inner iterations, progress and safety figures.  `warm`: previous plan shifted by one step as initial guess
    (the reference passes initial_guess=None, i.e. cold: src/interface_mpc.py:82)."""
    import torch
    from trajtrack_mpcndqn_rlboost_amd import BatchSolver
    from trajtrack_mpcndqn_rlboost_amd.device_tracker import DeviceTracker
    from trajtrack_mpcndqn_rlboost_amd.feeders import DYN_OBS_SIZE
    N = int(cfg.N_hor)
    dev = torch.device("cuda", device)
    solver = BatchSolver(cfg, device=device, order=order, tail_promotion=tail_promotion)
    dt = DeviceTracker(cfg, B, device=device, solver=solver)
    y0, pos_h, vel_h, static = scene_one(B, n_dyn, seed)
    setup(dt, B, y0, static)
    pos = torch.from_numpy(pos_h).to(dev)
    vel = torch.from_numpy(vel_h).to(dev)
    k = torch.arange(1, N + 1, dtype=torch.float64, device=dev)
    pred = torch.zeros(B, n_dyn, N, 6, dtype=torch.float64, device=dev)
    pred[..., 2] = DYN_OBS_SIZE; pred[..., 3] = DYN_OBS_SIZE; pred[..., 5] = 1.0

    def predictions():      # est_dyn_obs_positions (src/main.py:77-85): current + (k + 1) * (current - last)
        delta = pos - (pos - vel)
        pred[..., 0] = pos[..., None, 0] + delta[..., None, 0] * k
        pred[..., 1] = pos[..., None, 1] + delta[..., None, 1] * k
        return pred
    total = warmup_ticks + ticks
    statuses = torch.zeros(total, B, dtype=torch.int32, device=dev)     # histogram after the loop: torch.bincount reads its bounds back
    inner = torch.zeros(total, dtype=torch.float64, device=dev)
    inside_box = torch.zeros(B, dtype=torch.bool, device=dev)
    hit_disc = torch.zeros(B, dtype=torch.bool, device=dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(total + 1)]
    guess = None
    ordered = []
    for t in range(total):
        ev[t].record()
        dt.set_dynamic_constraints(predictions())
        out = dt.step(initial_guess=guess)
        if t == 0:           # the first tick found the batch's shape (count read-back); the following ticks read nothing back
            torch.cuda.synchronize()
            sh = solver.last_shape()
            solver.reserve_shape(max_static=sh["max_static"], max_fleet=sh["max_fleet"], max_dyn=sh["max_dyn"],
                                 var_shape=not sh["shape_const"], axis_aligned=sh["axis_aligned"])
            solver.reserve_batch(B)
        ordered.append(bool(solver.last_shape()["ordered"]))
        statuses[t].copy_(out["status"])
        inner[t] = out["inner_it"].to(torch.float64).mean()
        if warm:
            u = out["u"].view(B, N, 2)
            guess = torch.cat([u[:, 1:], u[:, -1:]], dim=1).reshape(B, 2 * N).contiguous()
        pos += vel
        x, y = dt.states[:, 0], dt.states[:, 1]
        inside_box |= (x > 7.5) & (x < 8.5) & (y > 3.0) & (y < 4.0)
        hit_disc |= (torch.hypot(pos[..., 0] - x[:, None], pos[..., 1] - y[:, None]) < 0.8).any(dim=1)
    ev[total].record()
    torch.cuda.synchronize()
    tick_ms = [ev[t].elapsed_time(ev[t + 1]) for t in range(total)]
    timed = tick_ms[warmup_ticks:]
    hist_h = np.stack([np.bincount(row, minlength=5)[:5] for row in statuses.cpu().numpy()])   # all five codes (3 not finite, 4 shape exceeded)
    if hist_h[:, 3:].sum():
        raise RuntimeError(f"closed loop: {hist_h[:, 3].sum()} non-finite and {hist_h[:, 4].sum()} shape-exceeded solves (reservation of tick 0 broken?)")
    res = {"batch": B, "ticks": ticks, "warmup_ticks": warmup_ticks, "n_dyn": n_dyn, "start": "warm" if warm else "cold",
           "order": order, "ordered_ticks": int(sum(ordered[warmup_ticks:])),
           "ms_per_tick": float(np.mean(timed)), "ms_per_tick_min_max": [float(min(timed)), float(max(timed))],
           "ms_of_every_tick": [round(float(x), 2) for x in timed],
           "value": B * ticks / (sum(timed) * 1e-3), "unit": "solves/s",
           "status_histogram_per_tick": hist_h[warmup_ticks:, :3].tolist(),   # (codes 3 and 4 are zero: checked above)
           "status_histogram_total": hist_h[warmup_ticks:, :3].sum(axis=0).tolist(),
           "converged_fraction": float(hist_h[warmup_ticks:, 0].sum() / (B * ticks)),
           "mean_inner_iterations": float(inner[warmup_ticks:].mean()),
           "mean_x_after": float(dt.states[:, 0].mean()), "still_active": float(dt.active.to(torch.float64).mean()),
           "entered_box": float(inside_box.to(torch.float64).mean()), "touched_disc": float(hit_disc.to(torch.float64).mean()),
           "lds_bytes_per_wavefront": solver.last_shape()["lds_bytes"], "wavefronts_per_simd": solver.last_shape()["waves_per_simd"]}
    res["_final_states"] = dt.states.cpu().numpy()
    solver.close()
    return res


def device_closed_loop_streams(cfg, B=8192, streams=2, ticks=30, warmup_ticks=5, n_dyn=4, warm=False, order="longest_first", device=0, seed=5):
    """The same loop with the fleet split into `streams` sub-fleets, each with its own DeviceTracker, solver handle and HIP stream,
    ticking back to back without waiting for one another: the launch of one sub-fleet fills the compute units that the tail of
    another one is draining (the robots of a tick are independent: src/scenario_simulator.py:226-233 couples them only through the
    PREVIOUS tick's predictions).  What matters for a controller that has to act every `ts`: the PERIOD of a sub-fleet's ticks
    (start to start on its stream = its own tick duration under the others' load).  Returns per-tick periods of every sub-fleet."""
    import torch
    from trajtrack_mpcndqn_rlboost_amd import BatchSolver
    from trajtrack_mpcndqn_rlboost_amd.device_tracker import DeviceTracker
    from trajtrack_mpcndqn_rlboost_amd.feeders import DYN_OBS_SIZE
    N = int(cfg.N_hor)
    dev = torch.device("cuda", device)
    sizes = [B // streams + (1 if i < B % streams else 0) for i in range(streams)]
    total = warmup_ticks + ticks
    k = torch.arange(1, N + 1, dtype=torch.float64, device=dev)
    fleets = []
    for i, b in enumerate(sizes):
        solver = BatchSolver(cfg, device=device, order=order)
        dt = DeviceTracker(cfg, b, device=device, solver=solver)
        y0, pos_h, vel_h, static = scene_one(b, n_dyn, seed + 101 * i)
        setup(dt, b, y0, static)
        pred = torch.zeros(b, n_dyn, N, 6, dtype=torch.float64, device=dev)
        pred[..., 2] = DYN_OBS_SIZE; pred[..., 3] = DYN_OBS_SIZE; pred[..., 5] = 1.0
        fleets.append(dict(solver=solver, dt=dt, pos=torch.from_numpy(pos_h).to(dev), vel=torch.from_numpy(vel_h).to(dev), pred=pred, b=b,
                           stream=torch.cuda.Stream(device=dev), guess=None,
                           ev=[torch.cuda.Event(enable_timing=True) for _ in range(total + 1)],
                           statuses=torch.zeros(total, b, dtype=torch.int32, device=dev)))
    torch.cuda.synchronize()
    for t in range(total):
        for f in fleets:
            with torch.cuda.stream(f["stream"]):
                f["ev"][t].record()
                f["pred"][..., 0] = f["pos"][..., None, 0] + f["vel"][..., None, 0] * k
                f["pred"][..., 1] = f["pos"][..., None, 1] + f["vel"][..., None, 1] * k
                f["dt"].set_dynamic_constraints(f["pred"])
                out = f["dt"].step(initial_guess=f["guess"])
                if t == 0:
                    f["stream"].synchronize()
                    sh = f["solver"].last_shape()
                    f["solver"].reserve_shape(max_static=sh["max_static"], max_fleet=sh["max_fleet"], max_dyn=sh["max_dyn"],
                                              var_shape=not sh["shape_const"], axis_aligned=sh["axis_aligned"])
                    f["solver"].reserve_batch(f["b"])
                f["statuses"][t].copy_(out["status"])
                if warm:
                    u = out["u"].view(f["b"], N, 2)
                    f["guess"] = torch.cat([u[:, 1:], u[:, -1:]], dim=1).reshape(f["b"], 2 * N).contiguous()
                f["pos"] += f["vel"]
    for f in fleets:
        with torch.cuda.stream(f["stream"]):
            f["ev"][total].record()
    torch.cuda.synchronize()
    periods = [[f["ev"][t].elapsed_time(f["ev"][t + 1]) for t in range(warmup_ticks, total)] for f in fleets]
    span = max(fleets[0]["ev"][warmup_ticks].elapsed_time(f["ev"][total]) for f in fleets)
    hist = np.stack([np.bincount(np.concatenate([f["statuses"][t].cpu().numpy() for f in fleets]), minlength=5)[:5] for t in range(warmup_ticks, total)])
    if hist[:, 3:].sum():
        raise RuntimeError("closed loop (streams): non-finite or shape-exceeded solves")
    res = {"batch": B, "streams": streams, "sub_fleets": sizes, "ticks": ticks, "start": "warm" if warm else "cold", "order": order,
           "worst_period_ms": float(max(max(p) for p in periods)), "mean_period_ms": float(np.mean(periods)),
           "periods_ms": [[round(float(x), 2) for x in p] for p in periods],
           "value": B * ticks / (span * 1e-3), "unit": "solves/s", "converged_fraction": float(hist[:, 0].sum() / (B * ticks)),
           "status_histogram_last_tick": hist[-1, :3].tolist()}
    for f in fleets:
        f["solver"].close()
    return res


def realtime_capacity(cfg, warm=False, order="longest_first", lo=2048, hi=12288, step=512, ticks=30, warmup_ticks=5, n_dyn=4, device=0,
                      limit_ms=None, streams=1):
    """The largest fleet per GPU (a multiple of `step`) whose WORST control tick of the scene-1 run stays within the sampling
    time `ts` of the yaml (config/mpc_default.yaml: 0.2 s) -- what the reference prints per step as its solve time
    (src/main.py:230-238), asked of a fleet.  Bisection over device_closed_loop runs; returns the sizes tried with their worst /
    mean tick and the per-tick times of the run at the answer."""
    limit_ms = 1e3 * float(cfg.ts) if limit_ms is None else limit_ms
    tried = {}

    def worst(B):
        if B not in tried:
            if streams > 1:
                r = device_closed_loop_streams(cfg, B, streams, ticks, warmup_ticks, n_dyn, warm, order, device=device)
                r["ms_per_tick_min_max"] = [0.0, r["worst_period_ms"]]; r["ms_per_tick"] = r["mean_period_ms"]
                r["ms_of_every_tick"] = r["periods_ms"]
            else:
                r = device_closed_loop(cfg, B, ticks, warmup_ticks, n_dyn, warm, order, device=device)
            tried[B] = r
        return tried[B]["ms_per_tick_min_max"][1]
    a, b = lo // step, hi // step
    if worst(a * step) > limit_ms:
        best = 0
    elif worst(b * step) <= limit_ms:
        best = b * step
    else:
        while b - a > 1:
            m = (a + b) // 2
            if worst(m * step) <= limit_ms: a = m
            else: b = m
        best = a * step
    out = {"limit_ms": limit_ms, "start": "warm" if warm else "cold", "order": order, "robots": best, "step": step, "streams": streams,
           "tried": {str(B): {"worst_ms": round(r["ms_per_tick_min_max"][1], 2), "mean_ms": round(r["ms_per_tick"], 2)} for B, r in sorted(tried.items())}}
    if best:
        out["ms_of_every_tick"] = tried[best]["ms_of_every_tick"]
        out["converged_fraction"] = tried[best]["converged_fraction"]
    return out


def host_loop(B, T, K, warm):
    from trajtrack_mpcndqn_rlboost_amd import BatchedTracker, MpcConfig
    from trajtrack_mpcndqn_rlboost_amd.feeders import constant_velocity_prediction
    cfg = MpcConfig()
    bt = BatchedTracker(cfg, B, warm_start=warm)
    y0, pos, vel, static = scene_one(B, K)
    setup(bt, B, y0, static)
    inside_box = np.zeros(B, bool); hit_disc = np.zeros(B, bool)
    t0 = time.time(); solve_ms = []
    for t in range(T):
        pred = constant_velocity_prediction(pos - vel, pos, steps=cfg.N_hor)         # [B, K, N, 6]
        bt.set_dynamic_constraints(pred)
        bt.step("work")
        solve_ms.append(bt.solver.last_timing()["solve_ms"])
        pos = pos + vel
        x, y = bt.states[:, 0], bt.states[:, 1]
        inside_box |= (x > 7.5) & (x < 8.5) & (y > 3.0) & (y < 4.0)
        hit_disc |= (np.hypot(pos[..., 0] - x[:, None], pos[..., 1] - y[:, None]) < 0.8).any(axis=1)   # physical radius 0.8
    wall = time.time() - t0
    d_goal = np.hypot(bt.states[:, 0] - 15.4, bt.states[:, 1] - 3.5)
    print(f"[host assembly] B={B} ticks={T} discs={K} {'warm' if warm else 'cold'} start: wall {wall:.1f}s  kernel {np.mean(solve_ms):.1f} ms/tick  "
          f"({B * T / (np.sum(solve_ms) * 1e-3):.0f} solves/s in-kernel)")
    print(f"  progress: mean x {bt.states[:, 0].mean():.2f} m (start 0.6), within 0.5 m of goal: {(d_goal < 0.5).mean():.2f}, still active {bt.active.mean():.2f}")
    print(f"  safety  : entered the (un-inflated) box {inside_box.mean():.3f}, touched a disc (0.8 m) {hit_disc.mean():.3f}")
    print(f"  last tick status histogram {np.bincount(bt.last_result.status, minlength=4).tolist()} mean inner it {bt.last_result.num_inner_iterations.mean():.0f}")
    return bt.states.copy()


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 80
    K = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    WARM = "warm" in sys.argv[4:]
    if "capacity" in sys.argv[4:]:
        import json
        from trajtrack_mpcndqn_rlboost_amd import MpcConfig
        for streams, hi in ((1, 8192), (2, 12288)):
            for warm in (False, True):
                r = realtime_capacity(MpcConfig(), warm=warm, n_dyn=K, streams=streams, hi=hi)
                r.pop("ms_of_every_tick", None)
                print(json.dumps(r), flush=True)
    elif "streams" in sys.argv[4:]:
        from trajtrack_mpcndqn_rlboost_amd import MpcConfig
        for streams in (1, 2, 3):
            r = device_closed_loop_streams(MpcConfig(), B, streams, T, 5, K, WARM)
            print(f"[device loop] {streams} stream(s) x {r['sub_fleets']}: {r['value']:.0f} solves/s, tick period mean {r['mean_period_ms']:.1f} ms, worst {r['worst_period_ms']:.1f} ms, "
                  f"converged {r['converged_fraction']:.3f}, last tick {r['status_histogram_last_tick']}", flush=True)
    elif "host" in sys.argv[4:]:
        host_loop(B, T, K, WARM)
    else:
        from trajtrack_mpcndqn_rlboost_amd import MpcConfig
        for order in ("as_given", "longest_first"):
            r = device_closed_loop(MpcConfig(), B, T, 5, K, WARM, order)
            r.pop("_final_states")
            per_tick = r.pop("status_histogram_per_tick")
            print(f"[device loop] order {order:13s} {r['start']} start: {r['ms_per_tick']:.1f} ms/tick ({r['value']:.0f} solves/s), ordered ticks "
                  f"{r['ordered_ticks']}/{T}, status total {r['status_histogram_total']}, first / last tick {per_tick[0]} / {per_tick[-1]}, "
                  f"mean inner {r['mean_inner_iterations']:.0f}, mean x {r['mean_x_after']:.2f} m, entered box {r['entered_box']:.3f}, "
                  f"touched a disc {r['touched_disc']:.3f}")
