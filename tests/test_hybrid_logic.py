"""Hybrid DQN + MPC decision logic: the switcher, the reference filter and the obstacle
inflation against vectors produced by the reference's own definitions (tests/golden/make_hybrid_fixtures.py)."""
import importlib
import json
import math
import os

import numpy as np
import pytest

hybrid = importlib.import_module("trajtrack_mpcndqn_rlboost_amd.hybrid")
FX = np.load(os.path.join(os.path.dirname(__file__), "golden", "hybrid_switch.npz"))


@pytest.mark.parametrize("case", [0, 1, 2])
def test_hint_switcher_follows_the_reference_state_machine(case):
    boxes = json.loads(bytes(FX["boxes_json"]).decode())
    sd, dd, ds = FX[f"sw{case}_args"]
    sw = hybrid.HintSwitcher(sd, dd, ds)
    res, cnt = [], []
    for pos, orig, new, rect in zip(FX[f"sw{case}_pos"], FX[f"sw{case}_orig"], FX[f"sw{case}_new"], FX[f"sw{case}_rect"]):
        res.append(sw.switch(tuple(pos), orig.tolist(), new.tolist(), boxes + [rect.tolist()]))
        cnt.append(sw.detach_cnt)
    assert np.array_equal(np.array(res), FX[f"sw{case}_res"])
    assert np.array_equal(np.array(cnt), FX[f"sw{case}_cnt"])
    assert 0 < FX[f"sw{case}_res"].sum() < len(res)          # both states occur


def test_reference_filter_and_circle_to_rect():
    o, n = FX["filt_orig"], FX["filt_new"]
    for d in (1, 0.9, 0.5, 0.05):
        assert np.array_equal(hybrid.ref_traj_filter(o, n, decay=d), FX[f"filt_{d}"])
    assert np.array_equal(hybrid.ref_traj_filter(o, n, decay=1), n)   # decay 1 never decays: the proposal as it is
    assert np.array_equal(np.array(hybrid.circle_to_rect([3.0, -2.0])), FX["rect"])


def test_polygon_distance_and_mitre_inflation():
    sq = [(0, 0), (2, 0), (2, 2), (0, 2)]
    assert hybrid.polygon_distance(sq, (1, 1)) == 0.0
    assert hybrid.polygon_distance(sq, (3, 1)) == 1.0
    assert math.isclose(hybrid.polygon_distance(sq, (3, 3)), math.sqrt(2))
    # rectangle + 0.8 mitre = rectangle grown by 0.8 on every side (what the MPC sees: main.py:110)
    grown = np.array(hybrid.inflate_polygon(sq, 0.8))
    assert {tuple(np.round(p, 12)) for p in grown} == {(-0.8, -0.8), (2.8, -0.8), (2.8, 2.8), (-0.8, 2.8)}
    # every inflated edge is parallel to, and 0.8 away from, its source edge -- also for a general convex quadrilateral
    quad = [(4.0, 0.0), (4.0, 13.0), (4.5, 13.0), (10.0, 0.0)]       # the "sharp" block of scene 2 (map.py)
    out = np.array(hybrid.inflate_polygon(quad, 0.8))
    assert len(out) == 4
    src = hybrid.rg.orient(quad)
    for i in range(4):
        a, b = src[i], src[(i + 1) % 4]
        e = (b - a) / np.hypot(*(b - a))
        nrm = np.array([e[1], -e[0]])
        assert abs((out[i] - a) @ nrm - 0.8) < 1e-12 and abs((out[(i + 1) % 4] - a) @ nrm - 0.8) < 1e-12
    # a needle sharper than the mitre limit is bevelled at 5 x margin from the tip
    needle = [(0.0, 0.0), (10.0, 0.2), (10.0, -0.2)]
    out = np.array(hybrid.inflate_polygon(needle, 0.5))
    assert len(out) == 4 and np.isclose(np.min(out[:, 0]), -2.5, atol=1e-9)
