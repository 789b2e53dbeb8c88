"""Scripted multimodal scanner scenes of the reference (scenes 1-5, src/scenario_simulator.py:71-133).

tests/golden/scanner_scenes.npz holds what the reference's own map and obstacle classes return (generator:
tests/golden/make_scanner_fixture.py).
* CPU: `feeders.scanner_prediction` restates the re-ordering of obstacle_simulator/_obstacle_simulator.py:48-76 -- checked
  against the recorded input / output pairs of every scene; the rows fit the configured dynamic-obstacle slots.
* GPU: tools/scanner_replay.py drives a batch of worlds per scene through BatchedTracker (every mode of every obstacle one
  dynamic-obstacle row, rows born and dying over time, shapes changing along the horizon = the general tables; scene 5: two
  coupled robots per world, Gauss-Seidel like the reference's sequential loop)."""
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from trajtrack_mpcndqn_rlboost_amd import MpcConfig  # noqa: E402
from trajtrack_mpcndqn_rlboost_amd.feeders import scanner_prediction  # noqa: E402

FX = np.load(os.path.join(ROOT, "tests", "golden", "scanner_scenes.npz"))


@pytest.mark.parametrize("s", [1, 2, 3, 4, 5])
def test_scanner_prediction_reorders_like_the_reference(s):
    raw, rows, radius = FX[f"raw_{s}"], FX[f"rows_{s}"], float(FX[f"radius_{s}"])
    K, N, M, _ = raw.shape
    assert K >= 4 and N == 20
    for k in range(K):
        got = scanner_prediction(np.transpose(raw[k], (1, 0, 2)), radius, factor=1.0)      # [M, N, 6]
        assert np.array_equal(got, rows[k, :M]), (s, k)
    cfg = MpcConfig()
    assert FX[f"nrows_{s}"].max() <= cfg.Ndynobs and rows.shape[2] == cfg.N_hor
    # the scenes exercise what the general tables exist for: rows whose shape changes along the horizon, several modes
    if s in (1, 2):
        assert M >= 2 or FX[f"nrows_{s}"].max() > 5
    changing = np.any(rows[:, :, 1:, 2] != rows[:, :, :1, 2])
    assert changing or s in (3, 4, 5)


@pytest.mark.gpu
@pytest.mark.parametrize("s", [1, 2, 3, 4, 5])
def test_batch_replay_of_the_scanner_scene(s):
    replay = importlib.import_module("scanner_replay")
    B = 64
    o = replay.replay(s, B, jitter=0.05)
    print("\n" + replay.summary(o))
    assert o["arrived"].all()                                    # every robot of every world reaches its goal ...
    assert np.median(o["goal_distance"]) < 0.1
    assert o["ticks"] < 120                                       # ... within the recorded horizon of the scanner
    assert o["status_histogram"][2:].sum() == 0                   # no time-outs, no non-finite solves, no shape overflow
    # What the reference's OWN simulator run shows for the same quantities (tests/golden/fleet_trace.npz `ref_summary_<s>`:
    # Simulator.run unchanged, nominal start pose, oracle stand-in behind the plugin; scenes 2-5 -- scene 1 needs the visibility
    # planner around its box).  [ticks, robots, converged calls, closest approach to a hard ellipse, robot-robot distance, ...]
    ref = np.load(os.path.join(ROOT, "tests", "golden", "fleet_trace.npz"))
    if s in (2, 4):
        core_ref = float(ref[f"ref_summary_{s}"][3])
        print(f"    reference run: closest approach to a hard ellipse {core_ref:.3f}; batch replay median {np.median(o['min_core']):.3f}")
        assert o["min_core"].min() > 0.95                         # nobody enters a hard ellipse ...
        assert abs(np.median(o["min_core"]) - core_ref) < 0.15 * core_ref      # ... and the typical clearance is the reference run's
    if s == 1:
        assert o["min_core"].min() > 0.95
    if s == 3:
        # Head-on "crash" scene: the obstacle comes down the robot's own lane.  From the exactly symmetric nominal pose the
        # reference run itself ends up INSIDE the hard ellipse (closest approach 0.20: nothing breaks the symmetry until the
        # penalty is large); a few centimetres of lateral offset decide the side and the robots pass on the ellipse's edge.
        core_ref = float(ref["ref_summary_3"][3])
        print(f"    reference run (symmetric start): closest approach {core_ref:.3f}; batch replay (5 cm jitter) min {o['min_core'].min():.3f}")
        # (the hard ellipse holds when the penalty grows -- "both": 1.000; under "either" c stays at 10 while the acceleration
        #  constraints are inactive and the hinge lets the robots 19 % into it: min 0.81, profiles/r06_stall_rule.txt)
        from trajtrack_mpcndqn_rlboost_amd.config import SOLVER_DEFAULTS
        assert core_ref < 0.5 and o["min_core"].min() > (0.95 if SOLVER_DEFAULTS["solver_penalty_stall"] == "both" else 0.75)
    if s == 5:
        # Two robots swap lanes.  The fleet term of the reference is a soft hinge on W^2 - d^2 with W = vehicle_width = 0.5 m
        # between CENTRES (mpc_generator.py:105-108, 211-216): it only acts below 0.5 m.  The reference's own run passes at
        # 0.63 m; the jittered worlds scatter around that (measured 0.28 .. 0.8 m), bit for bit the sequential loop's answers
        # (tests/test_gpu_fleet.py).
        pair_ref = float(ref["ref_summary_5"][4])
        print(f"    reference run: robot-robot distance {pair_ref:.3f} m; batch replay median {np.median(o['min_pair']):.3f} m, min {o['min_pair'].min():.3f} m")
        assert abs(np.median(o["min_pair"]) - pair_ref) < 0.2
        assert o["min_pair"].min() > 0.2
