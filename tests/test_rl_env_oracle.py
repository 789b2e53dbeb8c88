"""CPU tests for SURVEY.md section 8 row f3 (batched DRL environment): the oracle against traces produced by the
reference's own environment code, the host-side polygon preparation, the record layout and the C-ABI exports."""
import ctypes as C
import importlib
import json
import math
import os

import numpy as np
import pytest

from oracle import rl_env_numpy as orc

rl_env = importlib.import_module("trajtrack_mpcndqn_rlboost_amd.rl_env")
rg = importlib.import_module("trajtrack_mpcndqn_rlboost_amd.rl_geometry")
solver_mod = importlib.import_module("trajtrack_mpcndqn_rlboost_amd.solver")

GOLD = os.path.join(os.path.dirname(__file__), "golden", "env_rays_traces.npz")


def load_traces():
    fx = np.load(GOLD)
    specs = json.loads(bytes(fx["specs_json"]).decode())
    maps = {name: rl_env.make_map(sp["boundary"], sp["static"], sp["dynamic"], sp["start"], sp["goal"], sp["path"])
            for name, sp in specs.items()}
    return fx, maps


class _Recording(orc.OracleRaysEnv):
    def reset(self):
        self.first = super().reset()
        return self.first


@pytest.mark.parametrize("key", ["scene1_r0", "scene1_r1", "lhall_r0", "lhall_r1"])
def test_oracle_reproduces_reference_environment_traces(key):
    fx, maps = load_traces()
    env = _Recording(maps[key.split("_")[0]], time_step=float(fx[key + "_ts"]))
    ints, exts, rews, dones, flags, states = [env.first["internal"]], [env.first["external"]], [], [], [], [env.state.copy()]
    for a in fx[key + "_actions"]:
        o, r, d, _ = env.step(int(a))
        ints.append(o["internal"]); exts.append(o["external"]); rews.append(r); dones.append(d)
        flags.append([env.collided_obstacle, env.collided_boundary, env.reached_goal]); states.append(env.state.copy())
    # robot states, flags and rewards are float64 logic: exact; observations pass through float32 in the reference
    assert np.array_equal(np.asarray(states), fx[key + "_state"])
    assert np.array_equal(np.asarray(flags), fx[key + "_flags"])
    assert np.array_equal(np.asarray(dones), fx[key + "_done"])
    assert np.abs(np.asarray(rews) - fx[key + "_reward"]).max() <= 1e-12
    assert np.abs(np.asarray(ints) - fx[key + "_internal"]).max() <= 1e-6
    assert np.abs(np.asarray(exts) - fx[key + "_external"]).max() <= 2e-6
    # every kind of event occurs somewhere in the traces
    assert fx[key + "_done"].any()
    # observe-only path (main.py:181-189)
    for st, ei, ee in zip(fx[key + "_teleport"], fx[key + "_tele_internal"], fx[key + "_tele_external"]):
        env.state[:] = st
        o, r, _, _ = env.step(None)
        assert r == 0.0
        assert np.abs(o["internal"] - ei).max() <= 1e-6 and np.abs(o["external"] - ee).max() <= 2e-6


def test_traces_cover_all_flags():
    fx, _ = load_traces()
    seen = np.zeros(3, dtype=bool)
    for key in ["scene1_r0", "scene1_r1", "lhall_r0", "lhall_r1"]:
        seen |= fx[key + "_flags"].any(axis=0)
    assert seen.all()


def test_round_buffer_of_a_rectangle_follows_the_geos_fillet_rule():
    ring = rg.buffer_polygon([(0, 0), (4, 0), (4, 2), (0, 2)], 0.5, quad_segs=4)
    assert len(ring) == 20                       # 4 corners x (4 chords -> 5 points)
    assert rg.signed_area(ring) > 0
    # every vertex is at distance 0.5 from the rectangle; arc points sit at multiples of pi/8 around the corner
    for p in ring:
        dx = max(0.0 - p[0], 0.0, p[0] - 4.0)
        dy = max(0.0 - p[1], 0.0, p[1] - 2.0)
        assert abs(math.hypot(dx, dy) - 0.5) < 1e-12
    corner = [p for p in ring if p[0] > 4 - 1e-12 and p[1] > 2 - 1e-12]
    ang = sorted(math.atan2(p[1] - 2, p[0] - 4) for p in corner)
    assert np.allclose(ang, np.arange(5) * math.pi / 8, atol=1e-12)
    # area: rectangle + 4 side strips + 4 x (4 chord triangles of the quarter disc)
    expect = 8 + 2 * 0.5 * (4 + 2) + 4 * 4 * 0.5 * 0.25 * math.sin(math.pi / 8)
    assert abs(rg.signed_area(ring) - expect) < 1e-12


def test_small_turn_gets_no_fillet_point_and_inward_buffer_rounds_reflex_corners():
    # regular 20-gon: exterior angle 18 deg < 22.5 deg -> int(18 / 22.5 + 0.5) = 1 chord: two points per corner
    nodes = rg.ellipse_nodes(0.8, 0.8, 20)
    assert len(rg.buffer_polygon(nodes, 0.5)) == 40
    # L-shaped hall shrunk by 0.5: five convex corners -> mitre points, the reflex corner (5, 5) -> a quarter arc
    hall = [(0, 0), (12, 0), (12, 5), (5, 5), (5, 12), (0, 12)]
    ring = rg.buffer_polygon(hall, -0.5)
    assert len(ring) == 5 + 5
    arc = [p for p in ring if abs(math.hypot(p[0] - 5, p[1] - 5) - 0.5) < 1e-12]
    assert len(arc) == 5
    assert all(rg.point_in_ring(p, np.asarray(hall, dtype=float)) for p in ring)
    assert {(round(p[0], 9), round(p[1], 9)) for p in ring} >= {(0.5, 0.5), (11.5, 0.5), (11.5, 4.5), (4.5, 11.5), (0.5, 11.5)}


def test_buffer_refuses_outlines_where_the_local_construction_fails():
    with pytest.raises(ValueError):
        rg.buffer_polygon([(0, 0), (10, 0), (10, 0.6), (0, 0.6)], -0.5)   # the shrunk hall vanishes
    with pytest.raises(ValueError):   # a slot narrower than twice the padding: the two offset walls cross
        rg.buffer_polygon([(0, 0), (6, 0), (6, 4), (3.3, 4), (3.3, 1), (2.7, 1), (2.7, 4), (0, 4)], 0.5)


def test_keyframe_animation_matches_hand_values():
    # periodic motion (obstacle.py:97-105): cosine easing, cycle 2 * pi / freq, there and back
    ob = rl_env.periodic_obstacle((0.0, 0.0), (2.0, 4.0), freq=0.5, rx=0.5, ry=0.5, angle=0.0, corners=12)
    T = math.pi / 0.5
    pose = lambda t: orc.keyframe_pose(ob["time_steps"], ob["keyframes"], ob["interp"], ob["offset"], t)
    assert np.allclose(pose(0.0)[:2], (0, 0)) and np.allclose(pose(T / 2)[:2], (1, 2)) and np.allclose(pose(T * 0.999999)[:2], (2, 4), atol=1e-4)
    assert np.allclose(pose(1.5 * T)[:2], (1, 2)) and np.allclose(pose(2 * T + 0.25 * T)[:2], pose(0.25 * T)[:2])
    assert np.isclose(pose(0.25 * T)[0], 2 * (1 - math.cos(math.pi / 4)) / 2)
    # the reference's key frames carry the last node angle, not the caller's angle (obstacle.py:196-199)
    assert np.isclose(pose(0.3)[2], 2 * math.pi * 11 / 12)


def test_path_project_and_interpolate_follow_shapely_semantics():
    path = np.array([(0, 0), (4, 0), (4, 3)], dtype=float)
    cum = orc.path_lengths(path)
    assert np.allclose(cum, [0, 4, 7])
    assert orc.path_project(path, cum, (1.0, 2.0)) == 1.0
    assert orc.path_project(path, cum, (5.0, -1.0)) == 4.0        # closest to the corner node: first segment wins
    assert orc.path_project(path, cum, (9.0, 9.0)) == 7.0
    assert orc.path_interpolate(path, cum, 5.5) == (4.0, 1.5)
    assert orc.path_interpolate(path, cum, -1.0) == (0.0, 0.0) and orc.path_interpolate(path, cum, 99.0) == (4.0, 3.0)


def test_record_layout_matches_the_library_and_env_symbols_are_exported():
    _, maps = load_traces()
    rec, maxima = rl_env.pack_records(list(maps.values()))
    lib = rl_env._bind(solver_mod.load_library())
    for name in rl_env.ENV_EXPORTS:
        assert hasattr(lib, name)
    params = rl_env._CParams(num_segments=8, corner_samples=3, time_step=0.2, **maxima, **rl_env.ROBOT)
    assert lib.mpcgpu_env_record_doubles(C.byref(params)) == rec.shape[1]
    bad = rl_env._CParams(num_segments=8, corner_samples=3, n_path_max=1, n_obst_max=0, n_kf_max=1, n_edge_max=4)
    assert lib.mpcgpu_env_record_doubles(C.byref(bad)) < 0 and b"invalid" in lib.mpcgpu_env_last_error()
    # header and edge table of the first record
    m = list(maps.values())[0]
    n_edges = len(m["boundary_padded"]) + sum(len(o["padded_nodes"]) for o in m["obstacles"])
    assert rec[0, 0] == len(m["path"]) and rec[0, 1] == len(m["obstacles"]) and rec[0, 2] == n_edges
    header = open(os.path.join(os.path.dirname(os.path.dirname(__file__)), "include", "mpcgpu_env.h")).read()
    for name in rl_env.ENV_EXPORTS:
        assert name + "(" in header


def test_round_buffer_properties_on_random_convex_polygons():
    """Every vertex of the grown ring lies at distance r from the polygon (arc points and offset-edge end points), the
    ring is simple and counter-clockwise, contains the polygon, and its area is between the mitred-free lower bound
    (area + r * perimeter) and the exact Minkowski sum (+ pi r^2)."""
    rng = np.random.default_rng(3)
    for _ in range(60):
        n = int(rng.integers(3, 9))
        ang = np.sort(rng.uniform(0, 2 * math.pi, n))
        if np.min(np.diff(np.concatenate([ang, [ang[0] + 2 * math.pi]]))) < 0.25:
            continue
        rad = rng.uniform(0.5, 3.0)
        poly = np.stack([rad * np.cos(ang), rad * np.sin(ang)], axis=1) + rng.uniform(-5, 5, 2)
        if rg.signed_area(poly) < 0.2:
            continue
        r = float(rng.uniform(0.1, 1.0))
        ring = rg.buffer_polygon(poly, r, quad_segs=int(rng.integers(1, 9)))
        assert rg.signed_area(ring) > 0 and rg.ring_is_simple(ring)
        for p in ring:
            d = min(orc.point_segment_distance(p, poly[k], poly[(k + 1) % n]) for k in range(n))
            assert abs(d - r) < 1e-9 and not rg.point_in_ring(p, poly)
        assert all(rg.point_in_ring(v, ring) for v in poly)
        per = sum(math.hypot(*(poly[(k + 1) % n] - poly[k])) for k in range(n))
        area = rg.signed_area(poly)
        assert area + r * per - 1e-9 <= rg.signed_area(ring) <= area + r * per + math.pi * r * r + 1e-9
