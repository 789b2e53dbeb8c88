#!/usr/bin/env python3
"""Golden vectors for the hybrid decision logic (HintSwitcher, reference filter).

Run in the build container only (needs /root/reference):   python tests/golden/make_hybrid_fixtures.py

``HintSwitcher`` (src/main_pre.py:27-52), ``ref_traj_filter`` and ``circle_to_rect`` (src/main.py:34-41,86-90) are
taken from the reference's files by parsing them and compiling ONLY those definitions (the modules themselves import
gym / stable-baselines3 / shapely at the top, none of which is installed), then executed on seeded random inputs.
shapely's ``Polygon.contains / distance`` and ``Point`` are the shim of make_env_fixtures.py extended with
``distance``.  Only data is written: hybrid_switch.npz.
"""
import ast
import json
import math
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)

import make_env_fixtures as shim  # noqa: E402
from oracle import rl_env_numpy as orc  # noqa: E402


class Polygon(shim.Polygon):
    def distance(self, pt):
        if self.contains(pt):
            return 0.0
        r = self._ccw
        return min(orc.point_segment_distance(pt.xy, r[k], r[(k + 1) % len(r)]) for k in range(len(r)))


def extract(path, names, namespace):
    tree = ast.parse(open(path).read())
    body = [n for n in tree.body if isinstance(n, (ast.ClassDef, ast.FunctionDef)) and n.name in names]
    assert len(body) == len(names), (path, names)
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), namespace)


if __name__ == "__main__":
    from typing import List, Tuple
    ns = {"Polygon": Polygon, "Point": shim.Point, "List": List, "Tuple": Tuple, "np": np, "DYN_OBS_SIZE": 0.8 + 0.8}
    extract(os.path.join(REF, "src/main_pre.py"), ["HintSwitcher"], ns)
    extract(os.path.join(REF, "src/main.py"), ["ref_traj_filter", "circle_to_rect"], ns)
    rng = np.random.default_rng(11)
    out = {}
    # ---- switcher: a robot driving past boxes; the original reference crosses them, random proposals
    boxes = [[(4.0, 1.0), (4.0, 3.0), (6.0, 3.0), (6.0, 1.0)], [(10.0, 2.5), (10.0, 4.5), (12.5, 4.5), (11.0, 2.0)],
             [(17.0, -1.0), (17.0, 0.5), (18.0, 0.5), (18.5, -1.0)]]
    runs = []
    for case, (sd, dd, ds) in enumerate([(10, 2, 10), (3.0, 1.0, 2), (1.5, 0.5, 0)]):
        sw = ns["HintSwitcher"](sd, dd, ds)
        pos, orig, new, res, cnt, rects = [], [], [], [], [], []
        x = 0.0
        for t in range(140):
            x += rng.uniform(0.05, 0.3)
            p = (x, 2.0 + 0.8 * math.sin(0.35 * t) + rng.normal(0, 0.1))
            o = np.stack([p[0] + 0.24 * np.arange(1, 21), np.full(20, 2.0 + rng.normal(0, 0.3)), np.zeros(20)], axis=1)
            n = o + np.array([0.0, rng.uniform(1.0, 2.5), 0.0])
            rects.append(ns["circle_to_rect"]([x + 3.0 * math.cos(0.1 * t), 6.0 + 4.0 * math.sin(0.1 * t)]))
            obst = boxes + [rects[-1]]
            r = sw.switch(p, o.tolist(), n.tolist(), obst)
            pos.append(p); orig.append(o); new.append(n); res.append(r); cnt.append(sw.detach_cnt)
        out[f"sw{case}_args"] = np.array([sd, dd, ds], dtype=float)
        out[f"sw{case}_pos"] = np.array(pos); out[f"sw{case}_orig"] = np.array(orig); out[f"sw{case}_new"] = np.array(new)
        out[f"sw{case}_rect"] = np.array(rects)
        out[f"sw{case}_res"] = np.array(res, dtype=bool); out[f"sw{case}_cnt"] = np.array(cnt)
        print("switcher case", case, "on fraction", np.mean(res), "toggles", int(np.abs(np.diff(np.array(res, dtype=int))).sum()))
    out["boxes_json"] = np.frombuffer(json.dumps(boxes).encode(), dtype=np.uint8)
    # ---- reference filter
    o = rng.normal(size=(20, 3)); n = rng.normal(size=(20, 3))
    out["filt_orig"], out["filt_new"] = o, n
    for d in (1, 0.9, 0.5, 0.05):
        out[f"filt_{d}"] = ns["ref_traj_filter"](o, n, decay=d)
    out["rect"] = np.array(ns["circle_to_rect"]([3.0, -2.0]))
    np.savez_compressed(os.path.join(HERE, "hybrid_switch.npz"), **out)
    print("wrote hybrid_switch.npz")


def make_metrics_fixture():
    """The reference's own Metrics class (src/main_pre.py:55-144) on seeded random trials -> hybrid_metrics.npz."""
    import statistics
    from typing import List, Tuple
    ns = {"Polygon": Polygon, "Point": shim.Point, "List": List, "Tuple": Tuple, "np": np, "math": math, "statistics": statistics}
    extract(os.path.join(REF, "src/main_pre.py"), ["Metrics"], ns)
    rng = np.random.default_rng(21)
    m = ns["Metrics"]("hyb")
    out = {}
    obstacles = [[(4.0, 3.1), (4.0, 5.0), (6.0, 5.0), (6.0, 3.1)], [(10.0, 0.9), (10.0, -1.5), (12.5, -1.5), (11.0, 1.0)]]
    for t in range(7):
        n = int(rng.integers(20, 60))
        times = rng.uniform(1.0, 30.0, n)
        acts = np.stack([np.clip(np.cumsum(rng.normal(0, 0.1, n + 1)) + 1.0, -0.5, 1.5), rng.normal(0, 0.2, n + 1)], axis=1)
        ref = np.stack([np.linspace(0, 15, 63), np.full(63, 2.0), np.zeros(63)], axis=1)
        traj = np.stack([np.linspace(0, 15 * rng.uniform(0.5, 1.0), n + 2), 2.0 + rng.normal(0, 0.4, n + 2)], axis=1)
        ok = bool(rng.random() < 0.6)
        m.add_trial_result(times.tolist(), ok, [tuple(a) for a in acts], [tuple(r) for r in ref], [tuple(p) for p in traj], obstacles)
        out[f"t{t}_times"], out[f"t{t}_acts"], out[f"t{t}_ref"], out[f"t{t}_traj"], out[f"t{t}_ok"] = times, acts, ref, traj, np.array(ok)
        tr = m.trial_list[-1]
        out[f"t{t}_expect"] = np.array(tr["computation_time"] + tr["deviation_distance"] + tr["smoothness"] + [tr["clearance"], tr["finish_time"]], dtype=float)
    avg = m.get_average(4)
    out["average"] = np.array(avg["computation_time"] + avg["deviation_distance"] + avg["smoothness"] + [avg["clearance"], avg["finish_time"], avg["success_rate"]], dtype=float)
    out["obstacles_json"] = np.frombuffer(json.dumps(obstacles).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "hybrid_metrics.npz"), **out)
    print("wrote hybrid_metrics.npz", avg)


if __name__ == "__main__":
    make_metrics_fixture()
