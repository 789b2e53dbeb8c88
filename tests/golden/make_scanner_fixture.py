#!/usr/bin/env python3
"""Scripted multimodal scanner scenes of the reference (scenes 1-5 of src/scenario_simulator.py:71-133) as data.

For every scene the reference's own classes are executed (maps: pkg_map/preset_maps/scene_maps.py; obstacles:
obstacle_simulator/*_dynamic_obstacles.py) and what they RETURN is stored:
  * `rows_<s>`   [T, R, N, 6]  `scanner.get_full_obstacle_list(kt*ts, factor=1.0)` for kt = 0..T-1: every mode of every
                               obstacle is one row of N predicted (x, y, rx, ry, angle, alpha); zero padded to R rows
  * `nrows_<s>`  [T]           how many rows the scanner returned at that tick
  * `raw_<s>`    [K, N, M, 6]  the first K ticks of the first obstacle's own record `obs_dict['pred_Tj'][m]` =
                               (alpha, x, y, sx, sy, angle) -- the INPUT of the re-ordering that feeders.scanner_prediction
                               restates (obstacle_simulator/_obstacle_simulator.py:48-76), with `radius_<s>`
  * `map_<s>`    JSON: boundary, static obstacle polygons, start poses, way points
Run in the build container only (needs /root/reference):   python tests/golden/make_scanner_fixture.py"""
import json
import math
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REF, "src"))

from pkg_map.preset_maps.scene_maps import return_crosswalk_map, return_crossing_map  # noqa: E402 (reference)
from obstacle_simulator.crosswalk_ped_dynamic_obstacles import CrosswalkPedObstacleSimulator  # noqa: E402
from obstacle_simulator.crossing_busy_dynamic_obstacles import CrossingObstacleScanner  # noqa: E402
from obstacle_simulator.crosswalk_crash_dynamic_obstacles import CrosswalkCrashObstacleSimulator  # noqa: E402
from obstacle_simulator.crosswalk_follow_dynamic_obstacles import CrosswalkFollowObstacleSimulator  # noqa: E402

TS, N, T = 0.2, 20, 120


def scene(idx):
    """scenario_simulator.py:71-133, minus the plotting and the shapely inflator."""
    if idx == 1:
        b, o, _ = return_crosswalk_map()
        return b, o, [(0.6, 3.5, 0.0)], [[(15.4, 3.5, 0.0)]], CrosswalkPedObstacleSimulator(TS, 0.2, birth_time=-1), 0.2
    if idx == 2:
        b, o, _, _ = return_crossing_map()
        return (b, o, [(7, 0.6, math.radians(90))], [[(7, 11.5, math.radians(90)), (7, 15.4, math.radians(90))]],
                CrossingObstacleScanner(TS, obstacle_radius_list=[0.2, 0.2, 0.2, 0.5, 0.5]), 0.2)
    if idx == 3:
        b, o, _ = return_crosswalk_map(False)
        return b, o, [(0.6, 3.5, 0.0)], [[(15.4, 3.5, 0.0)]], CrosswalkCrashObstacleSimulator(TS, 0.5, birth_time=0), 0.5
    if idx == 4:
        b, o, _ = return_crosswalk_map(False)
        return b, o, [(0.6, 3.5, 0.0)], [[(15.4, 3.5, 0.0)]], CrosswalkFollowObstacleSimulator(TS, 0.2, birth_time=-3), 0.2
    b, o, _ = return_crosswalk_map(False)
    return (b, o, [(0.6, 4.0, 0.0), (0.6, 3.0, 0.0)], [[(15.4, 3.0, math.radians(180))], [(15.4, 4.0, math.radians(180))]],
            CrosswalkCrashObstacleSimulator(TS, 0.2, birth_time=0), 0.2)


if __name__ == "__main__":
    out = {}
    for s in range(1, 6):
        boundary, obstacles, starts, waypoints, scanner, radius = scene(s)
        ticks = [scanner.get_full_obstacle_list(current_time=kt * TS, factor=1.0) for kt in range(T)]
        R = max(len(t) for t in ticks)
        rows = np.zeros((T, R, N, 6))
        for kt, t in enumerate(ticks):
            for r, row in enumerate(t):
                rows[kt, r] = np.array(row, dtype=float)
        out[f"rows_{s}"] = rows
        out[f"nrows_{s}"] = np.array([len(t) for t in ticks])
        first = scanner.obstacle_sims[0] if hasattr(scanner, "obstacle_sims") else scanner
        raw = []
        for kt in range(T):
            d = first.get_obs_dict(kt * TS)
            if d is None or len(raw) == 8:
                continue
            keys = list(d)
            M = first.num_mode
            raw.append([[d[keys[j + 1]][m] if m < len(d[keys[j + 1]]) else d[keys[j + 1]][0] for m in range(M)] for j in range(N)])
        out[f"raw_{s}"] = np.array(raw, dtype=float)
        out[f"radius_{s}"] = np.array(first.r)
        out[f"map_{s}"] = np.frombuffer(json.dumps(dict(boundary=[list(map(float, p)) for p in boundary],
                                                         static=[[list(map(float, p)) for p in poly] for poly in obstacles],
                                                         starts=[list(map(float, p)) for p in starts],
                                                         waypoints=[[list(map(float, p)) for p in w] for w in waypoints])).encode(), dtype=np.uint8)
        print(f"scene {s}: rows per tick {out[f'nrows_{s}'].min()}..{R}, {len(obstacles)} static polygons, {len(starts)} robot(s)")
    np.savez_compressed(os.path.join(HERE, "scanner_scenes.npz"), **out)
