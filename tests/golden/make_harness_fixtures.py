#!/usr/bin/env python3
"""Golden traces of the reference's OWN tracker harness (InterfaceMpc + TrajectoryGenerator), driven by a fake
solver module, for the "next" rows f1 of SURVEY.md section 8 (parameter assembly, reference-trajectory sampling,
speed-reference rule, post-solve rollouts).

Run in the build container only (needs /root/reference):   python tests/golden/make_harness_fixtures.py

The reference harness imports `opengen` at class-definition time and a solver module named
`config.optimizer_name` (src/mpc_traj_tracker/trajectory_generator.py:25-27,63-71); both are provided as stubs in
sys.modules.  The fake solver is a deterministic function of the parameter vector so that the closed loop moves:
    v_k = 0.8 * vref_k,   w_k = 0.3 * sin(0.7 k + theta_0)          (k = 0..N-1)
Only inputs and outputs of the reference code are stored (harness_traces.npz), no source.
"""
import math
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

og = types.ModuleType("opengen")
og.opengen = og
og.tcp = types.ModuleType("opengen.tcp")
og.tcp.OptimizerTcpManager = object
og.tcp.solver_status = types.SimpleNamespace(SolverStatus=object)
sys.modules["opengen"] = og
sys.modules["opengen.tcp"] = og.tcp
cs = types.ModuleType("casadi.casadi")
cs.SX = type("SX", (), {})
cas = types.ModuleType("casadi")
cas.casadi = cs
sys.modules["casadi"] = cas
sys.modules["casadi.casadi"] = cs

N = 20
CAPTURED = []


def fake_u(p):
    p = np.asarray(p, dtype=float)
    vref = p[18 + 3 * N:18 + 4 * N]
    k = np.arange(N)
    return np.stack([0.8 * vref, 0.3 * np.sin(0.7 * k + p[2])], axis=1).reshape(-1)


class _Sol:
    def __init__(self, u, p):
        self.solution = u.tolist()
        self.cost = float(np.sum(u * u))
        self.exit_status = "Converged"
        self.solve_time_ms = 1.0


class _FakeSolver:
    def run(self, p, initial_guess=None):
        CAPTURED.append(np.array(p, dtype=float))
        return _Sol(fake_u(p), p)


for name in ("navi_default",):
    m = types.ModuleType(name)
    m.solver = lambda: _FakeSolver()
    sys.modules[name] = m

sys.path.insert(0, os.path.join(REF, "src"))
from util.mpc_config import Configurator  # noqa: E402 (reference)
from interface_mpc import InterfaceMpc  # noqa: E402 (reference)
from util import utils_geo  # noqa: E402 (reference)


def est_dyn_obs_positions(last_pos, current_pos, steps=20, size=1.6):  # format of src/main.py:77-85
    d = [current_pos[0] - last_pos[0], current_pos[1] - last_pos[1]]
    return [[current_pos[0] + d[0] * (i + 1), current_pos[1] + d[1] * (i + 1), size, size, 0, 1] for i in range(steps)]


def run_scene(init, goal, path, static_polys, dyn_tracks, steps, mode="work"):
    cfg = Configurator(os.path.join(REF, "config", "mpc_default.yaml"), verbose=False)
    mpc = InterfaceMpc(cfg, motion_model=None)
    mpc.initialization(np.array(init, dtype=float), np.array(goal, dtype=float), path, mode)
    mpc.update_static_constraints(static_polys)
    out = dict(p=[], action=[], pred=[], ref=[], idx=[], state=[], cost=[])
    global_ref = mpc.ref_traj.numpy()
    for t in range(steps):
        dyn = [est_dyn_obs_positions(tr(t - 1), tr(t)) for tr in dyn_tracks]
        mpc.update_dynamic_constraints(dyn)
        ref, _ = mpc.get_local_ref_traj()
        CAPTURED.clear()
        res = mpc.get_action(ref, mode=mode)
        if res is None:
            break
        action, pred, cost = res
        out["p"].append(CAPTURED[0]); out["action"].append(action); out["pred"].append(np.array(pred))
        out["ref"].append(ref); out["idx"].append(mpc._traj_gen.idx_ref); out["state"].append(mpc.state.copy())
        out["cost"].append(cost)
    return {k: np.array(v) for k, v in out.items()}, global_ref


if __name__ == "__main__":
    walls = [[(0.0, 1.5), (0.0, 1.6), (9.0, 1.6), (9.0, 1.5)], [(0.0, 8.4), (0.0, 8.5), (9.0, 8.5), (9.0, 8.4)],
             [(11.0, 1.5), (11.0, 1.6), (16.0, 1.6), (16.0, 1.5)], [(11.0, 8.4), (11.0, 8.5), (16.0, 8.5), (16.0, 8.4)]]
    box = [(7.5, 3.0), (7.5, 4.0), (8.5, 4.0), (8.5, 3.0)]
    path = [(0.6, 3.5), (5.0, 3.5), (8.0, 5.5), (11.0, 3.5), (15.4, 3.5)]
    save = {}
    # scene A: static only, long enough to run into the tail-padded local reference and the goal speed rule
    a, gref_a = run_scene((0.6, 3.5, 0.0), (15.4, 3.5, 0.0), path, walls + [box], [], steps=120)
    # scene B: two moving discs, 'safe' mode
    tracks = [lambda t: (10.0 - 0.12 * t, 3.5 + 0.01 * t), lambda t: (6.0 + 0.05 * t, 8.0 - 0.1 * t)]
    b, gref_b = run_scene((0.6, 3.5, 0.3), (15.4, 3.5, 0.0), path, walls, tracks, steps=25, mode="safe")
    for tag, d, g in (("A", a, gref_a), ("B", b, gref_b)):
        for k, v in d.items():
            save[f"{tag}_{k}"] = v
        save[f"{tag}_global_ref"] = g
    save["path"] = np.array(path)
    save["static_polys"] = np.array(walls + [box], dtype=float)
    print("scene A steps", len(a["p"]), "scene B steps", len(b["p"]), "global ref rows", len(gref_a))
    np.savez_compressed(os.path.join(HERE, "harness_traces.npz"), **save)
