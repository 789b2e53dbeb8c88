#!/usr/bin/env python3
"""Batch replay of the reference's scripted scanner scenes 1-5 (src/scenario_simulator.py:71-133, 211-250): B copies of a
scene's robot(s), start poses jittered, are driven through `BatchedTracker` -- ONE batched solve per tick (scene 5: one per
colour of the two-robot worlds, Gauss-Seidel as the reference's sequential loop) -- against the scene's static polygons and
the scanner's multimodal predictions (every mode of every obstacle is one dynamic-obstacle row; rows are born and die over
time and change shape along the horizon, i.e. the general obstacle tables).  The predictions come from
tests/golden/scanner_scenes.npz, recorded from the reference's own scanner classes.
usage: scanner_replay.py [B = 256] [scenes = 1,2,3,4,5]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from trajtrack_mpcndqn_rlboost_amd import BatchedTracker, BatchSolver, MpcConfig  # noqa: E402
from trajtrack_mpcndqn_rlboost_amd.hybrid import inflate_polygon  # noqa: E402

FIX = os.path.join(ROOT, "tests", "golden", "scanner_scenes.npz")


def load_scene(s):
    fx = np.load(FIX)
    m = json.loads(bytes(fx[f"map_{s}"]).decode())
    return m, fx[f"rows_{s}"], fx[f"nrows_{s}"]


def replay(s: int, B: int, cfg=None, solver=None, seed: int = 0, max_ticks: int = 120, jitter: float = 0.25):
    """Returns a dict of per-robot outcomes and per-tick timings for scene `s` with B worlds."""
    cfg = cfg if cfg is not None else MpcConfig()
    m, rows, nrows = load_scene(s)
    R = len(m["starts"])                                           # robots per world (scene 5: two)
    margin = cfg.vehicle_width + cfg.vehicle_margin                # test_block_mpc.py:39
    static = [inflate_polygon(poly, margin) for poly in m["static"]]
    rng = np.random.default_rng(seed + s)
    bt = BatchedTracker(cfg, B * R, solver=solver)
    groups = [[w * R + r for r in range(R)] for w in range(B)]
    goals = np.zeros((B * R, 2))
    for w in range(B):
        for r in range(R):
            x0, y0, th0 = m["starts"][r]
            across = rng.uniform(-jitter, jitter) if jitter > 0 else 0.0                      # jitter across the driving direction
            start = np.array([x0 - across * np.sin(th0), y0 + across * np.cos(th0), th0])
            wp = m["waypoints"][r]
            path = [tuple(start[:2])] + [tuple(p[:2]) for p in wp]
            i = w * R + r
            bt.initialization(i, start, np.array(wp[-1], dtype=float), path, "work")
            bt.update_static_constraints(i, static)
            goals[i] = wp[-1][:2]
    N = cfg.N_hor
    min_core = np.full(B * R, np.inf)        # smallest normalised distance to an obstacle's HARD ellipse (1 = on its edge)
    min_pair = np.full(B, np.inf)            # scene 5: distance between the two robots of a world
    tick_ms, kern_ms, statuses = [], [], np.zeros(5, dtype=np.int64)
    arrived_at = np.full(B * R, -1)
    for t in range(max_ticks):
        if not bt.active.any():
            break
        k = int(nrows[min(t, len(nrows) - 1)])
        row = rows[min(t, len(rows) - 1), :k]                      # [k, N, 6]
        bt.dyn_constraints[:] = 0.0                                # scenario_simulator.py:221: a fresh block every tick
        if k:
            bt.set_dynamic_constraints(np.broadcast_to(row[None], (B * R, k, N, 6)))
        t0 = time.perf_counter()
        bt.step("work", groups=groups if R > 1 else None)
        tick_ms.append(1e3 * (time.perf_counter() - t0))
        kern_ms.append(bt.solver.last_timing()["solve_ms"])
        statuses += np.bincount(bt.last_result.status, minlength=5)[:5]
        # where the obstacles ARE after this tick = first predicted step of the tick's rows
        if k:
            ox, oy, rx, ry, ang = (row[:, 0, j] for j in range(5))
            dx, dy = bt.states[:, None, 0] - ox[None], bt.states[:, None, 1] - oy[None]
            a = dx * np.cos(ang)[None] + dy * np.sin(ang)[None]
            b_ = dx * np.sin(ang)[None] - dy * np.cos(ang)[None]
            d = np.sqrt((a / np.maximum(rx, 1e-9)[None]) ** 2 + (b_ / np.maximum(ry, 1e-9)[None]) ** 2)
            live = bt.active[:, None] & (row[:, 0, 5] > 0)[None]
            min_core = np.minimum(min_core, np.where(live, d, np.inf).min(axis=1))
        if R == 2:
            st = bt.states.reshape(B, 2, 3)
            both = bt.active.reshape(B, 2).all(axis=1)
            min_pair = np.minimum(min_pair, np.where(both, np.hypot(st[:, 0, 0] - st[:, 1, 0], st[:, 0, 1] - st[:, 1, 1]), np.inf))
        arrived_at[(arrived_at < 0) & ~bt.active] = t
    d_goal = np.hypot(bt.states[:, 0] - goals[:, 0], bt.states[:, 1] - goals[:, 1])
    return dict(scene=s, worlds=B, robots=B * R, ticks=len(tick_ms), arrived=(~bt.active), goal_distance=d_goal, min_core=min_core,
                min_pair=min_pair, tick_ms=np.array(tick_ms), kernel_ms=np.array(kern_ms), status_histogram=statuses, arrived_at=arrived_at,
                max_rows=int(nrows.max()), general_tables=bool(bt.solver.last_shape()["lds_bytes"] > 0))


def summary(o):
    s = (f"scene {o['scene']}: {o['worlds']} worlds x {o['robots'] // o['worlds']} robot(s), up to {o['max_rows']} obstacle rows; {o['ticks']} ticks, "
         f"{np.median(o['tick_ms']):.1f} ms/tick (solve kernel {np.median(o['kernel_ms']):.1f}); reached the goal {o['arrived'].mean():.3f} "
         f"(median tick {int(np.median(o['arrived_at'][o['arrived_at'] >= 0])) if (o['arrived_at'] >= 0).any() else -1}), final goal distance "
         f"median {np.median(o['goal_distance']):.2f} m; closest approach to an obstacle's hard ellipse (1 = edge): min "
         f"{o['min_core'].min():.2f}, median {np.median(o['min_core'][np.isfinite(o['min_core'])]) if np.isfinite(o['min_core']).any() else float('nan'):.2f}; "
         f"solver statuses {o['status_histogram'].tolist()}")
    if np.isfinite(o["min_pair"]).any():
        s += f"; robot-robot distance in a world: min {o['min_pair'].min():.2f} m"
    return s


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    scenes_ = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 3, 4, 5]
    jit = float(sys.argv[3]) if len(sys.argv) > 3 else 0.25
    cfg = MpcConfig()
    solver = BatchSolver(cfg)
    for s in scenes_:
        print(summary(replay(s, B, cfg, solver, jitter=jit)), flush=True)
