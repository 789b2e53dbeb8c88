"""GPU (-m gpu): the two-problems-per-wavefront lane model (`Duo`, MPCGPU_OPT_PAIRING = 1 / BatchSolver(pairing=2); DESIGN.md
section 7) -- an explicit option, not the product rule (it measured slower), but the same solver source instantiated on another
lane model, so it is held to the same bar: reference-derived cost / gradient fixtures, the oracle on scenes, converged solves
within the north-star tolerance, and independence from the problem that shares the wavefront."""
import numpy as np
import pytest

import oracle
from conftest import load_golden, make_cfg, oracle_cfg
from trajtrack_mpcndqn_rlboost_amd import BatchSolver, MpcGpuError, scenes

pytestmark = pytest.mark.gpu
RTOL = 1e-11


def _rel(a, b):
    a, b = np.atleast_1d(np.asarray(a, float)), np.atleast_1d(np.asarray(b, float))
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.max(np.abs(b), axis=-1, keepdims=True))))


def test_cost_and_gradient_against_the_reference_fixtures_and_the_oracle():
    cfg = make_cfg(20)
    bs = BatchSolver(cfg, pairing=2)
    fx = load_golden("costgrad_N20.npz")
    r = bs.cost_grad(fx["u"], fx["p"], fx["c"], fx["y"])
    assert bs.last_shape()["problems_per_wavefront"] == 2
    for k, ref in (("f", "f"), ("psi", "psi"), ("grad", "grad_psi"), ("F1", "F1"), ("F2", "F2")):
        assert _rel(r[k], fx[ref]) < RTOL, k
    ocfg = oracle_cfg(cfg)
    B = 65                                                           # odd: the last wavefront carries one problem
    sc = scenes.make_batch(cfg, B, n_dyn=8, n_other=3, seed=120)
    rng = np.random.default_rng(20)
    u = np.stack([rng.uniform(-0.7, 1.8, (B, 20)), rng.uniform(-0.7, 0.7, (B, 20))], axis=2).reshape(B, 40)
    c = rng.choice([0.0, 10.0, 250.0, 6250.0], B)
    y = rng.uniform(-3, 3, (B, 40))
    r = bs.cost_grad(u, sc["p"], c, y)
    for i in range(B):
        o = oracle.cost_grad(ocfg, u[i], sc["p"][i], float(c[i]), y[i])
        for k in ("psi", "f", "grad", "F1", "F2"):
            assert _rel(r[k][i], o[k]) < RTOL, (i, k)
    bs.close()


def test_solves_match_the_oracle_and_the_one_per_wavefront_layout():
    cfg = make_cfg(20)
    B = 257
    sc = scenes.make_batch(cfg, B, n_dyn=8, seed=4321, dyn_clearance=0.1, box_clearance=0.3)
    duo, solo = BatchSolver(cfg, pairing=2), BatchSolver(cfg, pairing=1, latency_batch=0)
    a, b = duo.solve(sc["p"]), solo.solve(sc["p"])
    assert duo.last_shape()["problems_per_wavefront"] == 2 and solo.last_shape()["problems_per_wavefront"] == 1
    uo, _, ro, _ = oracle.solve_batch(oracle_cfg(cfg), sc["p"][:96])
    both = (a.status[:96] == 0) & (ro["status"] == 0)
    du = np.max(np.abs(a.solution[:96] - uo), axis=1)
    print(f"\n[Duo] converged on both sides {both.sum()}/96: |du|inf max {du[both].max():.2e}; against the one-per-wavefront layout: "
          f"same status {np.mean(a.status == b.status):.3f}, converged in both {((a.status == 0) & (b.status == 0)).sum()}")
    assert both.sum() >= 24 and du[both].max() <= 1e-3
    same = (a.status == 0) & (b.status == 0)
    assert same.sum() >= 0.3 * B and np.max(np.abs(a.solution - b.solution), axis=1)[same].max() <= 1e-3
    assert np.mean(a.status == b.status) > 0.9
    # five PANOC iterations from a non-zero guess track the oracle as tightly as the other layout does
    cfg5 = make_cfg(20, solver_max_inner_iterations=5, solver_max_outer_iterations=1)
    d5 = BatchSolver(cfg5, pairing=2)
    u0 = np.tile([0.6, 0.1], (96, 20))
    r5 = d5.solve(sc["p"][:96], u0)
    uo5, _, ro5, _ = oracle.solve_batch(oracle_cfg(cfg5), sc["p"][:96], u0)
    assert np.array_equal(r5.num_inner_iterations, ro5["inner_iters"]) and np.max(np.abs(r5.solution - uo5)) < 1e-6
    duo.close(); solo.close(); d5.close()


def test_a_problem_does_not_depend_on_the_problem_that_shares_its_wavefront():
    cfg = make_cfg(20)
    hard = scenes.make_batch(cfg, 40, n_dyn=8, seed=61)
    light = scenes.make_batch(cfg, 40, n_dyn=2, seed=62, dyn_clearance=0.1, box_clearance=0.3)
    bs = BatchSolver(cfg, pairing=2)
    a = bs.solve(hard["p"])
    # same problems, every one now paired with a different (lighter, differently shaped) neighbour and in the other half
    mix = np.empty((80, hard["p"].shape[1]))
    mix[1::2] = hard["p"]; mix[0::2] = light["p"]
    m = bs.solve(mix)
    assert np.array_equal(m.solution[1::2], a.solution) and np.array_equal(m.cost[1::2], a.cost)
    assert np.array_equal(m.num_inner_iterations[1::2], a.num_inner_iterations)
    alone = bs.solve(hard["p"][:1])                                  # one problem: the other half of the wavefront is empty
    assert np.array_equal(alone.solution[0], a.solution[0])
    bs.close()
    with pytest.raises(MpcGpuError, match="N_hor = 20"):
        BatchSolver(make_cfg(40), pairing=2)
