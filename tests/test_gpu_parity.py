"""GPU (-m gpu): the HIP path, called through the C-ABI, against the oracle and the golden fixtures.

Tolerances (float64 on both sides; they differ in summation order, FMA contraction and libm):
  * cost / gradient / constraint mappings:  relative 1e-11 against the reference-derived fixtures
  * solver iterates after k <= 20 PANOC iterations:  |du|_inf <= 1e-7 (same algorithm, rounding-level drift)
  * converged full solves:  |du|_inf <= 1e-3  -- the tolerance stated in BASELINE.json's north_star
  * non-converged full solves (the penalty method stops at an iteration cap; the iteration is chaotic:
    a 1-ulp input perturbation of the ORACLE ITSELF moves its answer by ~1e-2): compared statistically
    against that intrinsic sensitivity, plus solution-quality invariants.
"""
import numpy as np
import pytest

import oracle
from conftest import load_golden, make_cfg, oracle_cfg
from trajtrack_mpcndqn_rlboost_amd import BatchSolver, MpcGpuError, scenes, Solver

pytestmark = pytest.mark.gpu
RTOL_COST = 1e-11
U_TOL = 1e-3  # north_star tolerance on control sequences


def _rel(a, b):
    """max-norm error relative to the max-norm of the expected vector (per row for 2-D input)."""
    a, b = np.atleast_1d(np.asarray(a, float)), np.atleast_1d(np.asarray(b, float))
    scale = np.maximum(1.0, np.max(np.abs(b), axis=-1, keepdims=True))
    return float(np.max(np.abs(a - b) / scale))


def _rel_each(a, b):
    """element-wise relative error of scalars-per-problem arrays (cost, psi ...)."""
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))))


@pytest.mark.parametrize("N", [20, 40])
def test_cost_grad_matches_reference_fixtures(N, request):
    bs = request.getfixturevalue(f"solver{N}")
    fx = load_golden(f"costgrad_N{N}.npz")
    r = bs.cost_grad(fx["u"], fx["p"], fx["c"], fx["y"])
    assert _rel_each(r["f"], fx["f"]) < RTOL_COST
    assert _rel_each(r["psi"], fx["psi"]) < RTOL_COST
    assert _rel(r["grad"], fx["grad_psi"]) < RTOL_COST
    assert _rel(r["F1"], fx["F1"]) < RTOL_COST
    assert _rel(r["F2"], fx["F2"]) < RTOL_COST
    r0 = bs.cost_grad(fx["u"], fx["p"])                  # c = 0: psi is f, gradient is grad f
    assert _rel_each(r0["psi"], fx["f"]) < RTOL_COST
    assert _rel(r0["grad"], fx["grad_f"]) < RTOL_COST
    assert bs.last_shape()["max_dyn"] == bs.config.Ndynobs  # the dense fixtures fill every slot


@pytest.mark.parametrize("N,n_dyn,n_other", [(20, 8, 0), (20, 15, 10), (20, 0, 3), (40, 8, 2), (12, 3, 1)])
def test_cost_grad_matches_oracle_on_scenes(N, n_dyn, n_other):
    cfg = make_cfg(N)
    ocfg = oracle_cfg(cfg)
    bs = BatchSolver(cfg)
    B = 96
    sc = scenes.make_batch(cfg, B, n_dyn=n_dyn, n_other=n_other, seed=100 + N)
    rng = np.random.default_rng(N)
    u = np.stack([rng.uniform(-0.7, 1.8, (B, N)), rng.uniform(-0.7, 0.7, (B, N))], axis=2).reshape(B, 2 * N)
    c = rng.choice([0.0, 10.0, 250.0, 6250.0], B)
    y = rng.uniform(-3, 3, (B, 2 * N))
    r = bs.cost_grad(u, sc["p"], c, y)
    for i in range(B):
        o = oracle.cost_grad(ocfg, u[i], sc["p"][i], float(c[i]), y[i])
        assert _rel(r["psi"][i], o["psi"]) < RTOL_COST
        assert _rel(r["f"][i], o["f"]) < RTOL_COST
        assert _rel(r["grad"][i], o["grad"]) < RTOL_COST
        assert _rel(r["F1"][i], o["F1"]) < RTOL_COST
        assert _rel(r["F2"][i], o["F2"]) < RTOL_COST
    bs.close()


def test_zero_padding_keeps_reference_semantics_near_origin():
    """Padded other-robot rows sit at (0,0) and repel; padded dynamic rows are degenerate ellipses at (0,0)
    (mpc_generator.py:43,105-108,216).  Start the robot next to the origin so that both matter."""
    cfg = make_cfg(20)
    ocfg = oracle_cfg(cfg)
    bs = BatchSolver(cfg)
    sc = scenes.make_batch(cfg, 8, n_dyn=2, n_other=1, with_walls=False, with_box=False, seed=9)
    p = sc["p"].copy()
    p[:, 0] = np.linspace(-0.3, 0.3, 8); p[:, 1] = 1e-7; p[:, 2] = 0.0
    u = np.zeros((8, 40)); u[:, 0::2] = 0.05
    r = bs.cost_grad(u, p, np.full(8, 10.0))
    hit_fleet = 0
    for i in range(8):
        o = oracle.cost_grad(ocfg, u[i], p[i], 10.0)
        assert _rel(r["psi"][i], o["psi"]) < RTOL_COST and _rel(r["grad"][i], o["grad"]) < RTOL_COST
        assert _rel(r["F2"][i], o["F2"]) < RTOL_COST
        hit_fleet += o["f"] > 1000.0
    assert hit_fleet >= 4
    bs.close()


@pytest.mark.parametrize("k,tol", [(1, 1e-7), (2, 1e-7), (3, 1e-7), (5, 1e-6), (10, 1e-5), (20, 1e-3)])
def test_iterates_track_the_oracle_step_by_step(k, tol):
    """Same algorithm => after k PANOC iterations of the first inner problem the iterates agree, up to the
    exponential growth of rounding differences (measured: 1e-9 at k=1, 3e-7 at k=10, 1e-5 at k=20).
    A non-zero initial guess is used on purpose: with u0 = 0 the published local-Lipschitz estimate uses the
    perturbation h = 1e-12, whose gradient difference is at rounding-noise level, so gamma itself already
    differs by ~1e-4 between any two float64 implementations (see the cold-start test below)."""
    cfg = make_cfg(20, solver_max_inner_iterations=k, solver_max_outer_iterations=1)
    bs = BatchSolver(cfg)
    sc = scenes.make_batch(cfg, 64, n_dyn=8, seed=21)
    u0 = np.tile([0.6, 0.1], (64, 20))
    res = bs.solve(sc["p"], u0)
    uo, yo, ro, _ = oracle.solve_batch(oracle_cfg(cfg), sc["p"], u0)
    assert np.array_equal(res.num_inner_iterations, ro["inner_iters"])
    assert np.array_equal(res.status, ro["status"])
    assert np.max(np.abs(res.solution - uo)) < tol
    assert _rel_each(res.cost, ro["cost"]) < 10 * tol
    assert np.max(np.abs(res.lagrange_multipliers - yo)) < 100 * tol * max(1.0, np.max(np.abs(yo)))
    bs.close()


def test_cold_start_first_steps_track_within_lipschitz_estimate_noise():
    """u0 = 0 (what every reference call site uses, interface_mpc.py:82): h = 1e-12 makes the initial
    Lipschitz estimate noise-limited (relative ~1e-4), so early iterates agree to ~1e-4, not to rounding."""
    cfg = make_cfg(20, solver_max_inner_iterations=1, solver_max_outer_iterations=1)
    bs = BatchSolver(cfg)
    sc = scenes.make_batch(cfg, 64, n_dyn=8, seed=21)
    res = bs.solve(sc["p"])
    uo, _, ro, _ = oracle.solve_batch(oracle_cfg(cfg), sc["p"])
    du = np.max(np.abs(res.solution - uo), axis=1)
    assert np.array_equal(res.num_inner_iterations, ro["inner_iters"])
    assert np.median(du) < 1e-3 and du.max() < 5e-2
    bs.close()


def test_converged_solves_match_within_north_star_tolerance(solver20, cfg20):
    sc = scenes.make_batch(cfg20, 256, n_dyn=0, with_box=False, seed=31)
    res = solver20.solve(sc["p"])
    uo, yo, ro, _ = oracle.solve_batch(oracle_cfg(cfg20), sc["p"])
    both = (res.status == 0) & (ro["status"] == 0)
    assert both.sum() >= 64                                   # a meaningful number of converged problems
    du = np.max(np.abs(res.solution - uo), axis=1)
    assert du[both].max() <= U_TOL
    assert np.median(du[both]) <= 1e-5
    assert _rel_each(res.cost[both], ro["cost"][both]) <= 1e-6
    assert (res.status == ro["status"]).mean() >= 0.85
    # feasibility of every returned sequence w.r.t. the input box (the half step is always projected)
    uu = res.solution.reshape(256, 20, 2)
    assert uu[..., 0].min() >= cfg20.lin_vel_min - 1e-12 and uu[..., 0].max() <= cfg20.lin_vel_max + 1e-12
    assert np.abs(uu[..., 1]).max() <= cfg20.ang_vel_max + 1e-12
    # reported cost is f(u): cross-check through the oracle's cost function
    ocfg = oracle_cfg(cfg20)
    for i in range(0, 256, 37):
        assert _rel(res.cost[i], oracle.cost_grad(ocfg, res.solution[i], sc["p"][i])["f"]) < 1e-10


def test_hard_scenes_agree_within_intrinsic_sensitivity(solver20, cfg20):
    """Benchmark scene (8 dynamic discs, box on the path): most solves stop at the outer-iteration cap.
    GPU-vs-oracle distance must not exceed what a 1-ulp perturbation does to the oracle itself."""
    ocfg = oracle_cfg(cfg20)
    B = 192
    sc = scenes.make_batch(cfg20, B, n_dyn=8, seed=41)
    res = solver20.solve(sc["p"])
    uo, _, ro, _ = oracle.solve_batch(ocfg, sc["p"])
    p2 = sc["p"].copy(); p2[:, 0] *= (1 + 2.3e-16); p2[:, 1] *= (1 - 2.3e-16)
    uo2, _, ro2, _ = oracle.solve_batch(ocfg, p2)
    d_gpu = np.max(np.abs(res.solution - uo), axis=1)
    d_self = np.max(np.abs(uo2 - uo), axis=1)
    assert np.median(d_gpu) <= 3.0 * np.median(d_self) + 1e-6
    assert np.quantile(d_gpu, 0.9) <= 3.0 * np.quantile(d_self, 0.9) + 1e-6
    c_gpu = np.abs(res.cost - ro["cost"]) / np.maximum(1.0, ro["cost"])
    c_self = np.abs(ro2["cost"] - ro["cost"]) / np.maximum(1.0, ro["cost"])
    assert np.median(c_gpu) <= 3.0 * np.median(c_self) + 1e-9
    # solution quality is the same: cost and constraint violation distributions
    assert abs(np.mean(res.cost) / np.mean(ro["cost"]) - 1.0) < 0.05
    # (||F2|| of cap-limited solves spreads over three decades -- 1e-3 .. 4 -- and more so under the "either" stall rule, where the
    #  penalty stops growing for part of them: the MEDIAN of 192 such values is itself noisy; the distributions are compared in log space
    #  and against what the oracle's own 1-ulp twin shows)
    lg = lambda x: np.sort(np.log10(np.maximum(x, 1e-12)))
    w1 = float(np.mean(np.abs(lg(res.f2_norm) - lg(ro["f2_norm"]))))      # Wasserstein-1 distance of the two samples, in decades
    print(f"\n||F2|| distributions (log10): GPU quartiles {np.quantile(lg(res.f2_norm), [0.25, 0.5, 0.75]).round(2).tolist()}, oracle "
          f"{np.quantile(lg(ro['f2_norm']), [0.25, 0.5, 0.75]).round(2).tolist()}, distance {w1:.3f} decades")
    assert w1 <= 0.15
    assert (res.status == ro["status"]).mean() >= 0.9
    assert abs(res.num_inner_iterations.mean() / ro["inner_iters"].mean() - 1.0) < 0.1
    # work counters: the oracle counts the same evaluations
    n_psi, n_grad = solver20.last_eval_counts(B)
    assert abs(n_psi.mean() / ro["n_cost_evals"].mean() - 1.0) < 0.1 and abs(n_grad.mean() / ro["n_grad_evals"].mean() - 1.0) < 0.1
    assert np.all(n_grad <= n_psi) and np.all(n_psi >= res.num_inner_iterations)


def test_warm_start_multipliers_and_penalty_arguments(solver20, cfg20):
    ocfg = oracle_cfg(cfg20)
    sc = scenes.make_batch(cfg20, 32, n_dyn=0, with_box=False, seed=51)
    cold = solver20.solve(sc["p"])
    rng = np.random.default_rng(1)
    u0 = scenes.shifted_warm_start(cold.solution)
    y0 = cold.lagrange_multipliers
    c0 = rng.choice([1.0, 10.0, 50.0], 32)
    res = solver20.solve(sc["p"], u0, y0, c0)
    uo, yo, ro, _ = oracle.solve_batch(ocfg, sc["p"], u0, y0, c0)
    both = (res.status == 0) & (ro["status"] == 0)
    assert both.sum() >= 8
    assert np.max(np.abs(res.solution - uo), axis=1)[both].max() <= U_TOL
    # a warm start with the converged multipliers saves iterations when the cold solve spent outer iterations on GROWING the penalty
    # ("both"); under "either" the penalty of these obstacle-free problems never grows and the random initial penalties 1 / 10 / 50
    # decide instead (measured: mean 1244 warm against 1212 cold) -- no claim there
    if cfg20.solver_penalty_stall == "both":
        assert res.num_inner_iterations.mean() < cold.num_inner_iterations.mean()


def test_results_are_deterministic_and_independent_of_batch_composition(solver20, cfg20):
    sc = scenes.make_batch(cfg20, 48, n_dyn=8, seed=61)
    light = scenes.make_batch(cfg20, 48, n_dyn=2, seed=62)
    a = solver20.solve(sc["p"])
    b = solver20.solve(sc["p"])
    assert np.array_equal(a.solution, b.solution) and np.array_equal(a.cost, b.cost)
    assert np.array_equal(a.num_inner_iterations, b.num_inner_iterations)
    # a shard of the batch, reversed and mixed with lighter problems (different LDS carve): bitwise the same
    mix = np.concatenate([light["p"][:5], sc["p"][10:30][::-1], light["p"][5:9]])
    c = solver20.solve(mix)
    assert np.array_equal(c.solution[5:25], a.solution[10:30][::-1])
    assert np.array_equal(c.cost[5:25], a.cost[10:30][::-1])
    d = solver20.solve(light["p"][:9])
    assert np.array_equal(c.solution[:5], d.solution[:5]) and np.array_equal(c.solution[25:], d.solution[5:9])


def test_horizon_40_and_generic_horizon(solver40, cfg40):
    sc = scenes.make_batch(cfg40, 32, n_dyn=8, seed=71)
    cfgk = make_cfg(40, solver_max_inner_iterations=8, solver_max_outer_iterations=1)
    bs = BatchSolver(cfgk)
    u0 = np.tile([0.6, 0.1], (32, 40))
    res = bs.solve(sc["p"], u0)
    uo, _, ro, _ = oracle.solve_batch(oracle_cfg(cfgk), sc["p"], u0)
    assert np.max(np.abs(res.solution - uo)) < 1e-6 and np.array_equal(res.num_inner_iterations, ro["inner_iters"])
    bs.close()
    # full solve, N = 40, easy scene
    sc = scenes.make_batch(cfg40, 96, n_dyn=0, with_box=False, seed=72, v_init_range=(1.0, 1.2))
    res = solver40.solve(sc["p"])
    uo, _, ro, _ = oracle.solve_batch(oracle_cfg(cfg40), sc["p"])
    both = (res.status == 0) & (ro["status"] == 0)
    # whether the AKKT test is met within 500 inner iterations with 80 unknowns is at the mercy of rounding (the oracle against
    # itself with 1-ulp perturbed inputs agrees on 5 of 23): a small floor, the tolerance is what is asserted
    assert both.sum() >= 3, (int((res.status == 0).sum()), int((ro["status"] == 0).sum()), int(both.sum()))
    assert np.max(np.abs(res.solution - uo), axis=1)[both].max() <= U_TOL
    assert abs(int((res.status == 0).sum()) - int((ro["status"] == 0).sum())) <= 24
    # a horizon without a compiled specialisation goes through the generic kernel
    cfg12 = make_cfg(12, solver_max_inner_iterations=10, solver_max_outer_iterations=2)
    bs = BatchSolver(cfg12)
    sc = scenes.make_batch(cfg12, 16, n_dyn=3, seed=73)
    u0 = np.tile([0.6, 0.1], (16, 12))
    res = bs.solve(sc["p"], u0)
    uo, _, ro, _ = oracle.solve_batch(oracle_cfg(cfg12), sc["p"], u0)
    assert np.max(np.abs(res.solution - uo)) < 1e-5
    bs.close()


def test_time_varying_obstacle_shapes_take_the_general_tables(cfg20):
    """Scanner-style predictions change the semi-axes / angle along the horizon
    (obstacle_simulator/_obstacle_simulator.py:48-76): the batch then uses the general 9-doubles-per-item LDS
    tables instead of the shape-constant ones.  Both layouts must give the same numbers as the oracle."""
    ocfg = oracle_cfg(cfg20)
    off = cfg20.offsets()
    N = 20
    sc = scenes.make_batch(cfg20, 48, n_dyn=6, seed=111)
    p = sc["p"].copy()
    od = p[:, off["od"]:off["od"] + cfg20.Ndynobs * 6 * N].reshape(48, cfg20.Ndynobs, N, 6)
    k = np.arange(N)
    od[:, :6, :, 2] = 0.6 + 0.05 * k            # growing uncertainty ellipse
    od[:, :6, :, 3] = 0.4 + 0.03 * k
    od[:, :6, :, 4] = 0.3 + 0.02 * k
    od[:, :6, :, 5] = np.linspace(1.0, 0.3, N)
    rng = np.random.default_rng(5)
    u = np.stack([rng.uniform(-0.5, 1.5, (48, N)), rng.uniform(-0.5, 0.5, (48, N))], axis=2).reshape(48, 2 * N)
    bs = BatchSolver(cfg20)
    r = bs.cost_grad(u, p, np.full(48, 50.0))
    n_hit = 0
    for i in range(48):
        o = oracle.cost_grad(ocfg, u[i], p[i], 50.0)
        assert _rel(r["psi"][i], o["psi"]) < RTOL_COST and _rel(r["grad"][i], o["grad"]) < RTOL_COST
        assert _rel(r["F2"][i], o["F2"]) < RTOL_COST
        n_hit += o["F2"].max() > 0
    assert n_hit >= 5
    bs.close()
    cfgk = make_cfg(20, solver_max_inner_iterations=6, solver_max_outer_iterations=2)
    bs = BatchSolver(cfgk)
    u0 = np.tile([0.6, 0.1], (48, 20))
    res = bs.solve(p, u0)
    uo, _, ro, _ = oracle.solve_batch(oracle_cfg(cfgk), p, u0)
    assert np.array_equal(res.num_inner_iterations, ro["inner_iters"])
    assert np.max(np.abs(res.solution - uo)) < 1e-5
    # a mixed batch (some rows shape-constant, some not) and the pure shape-constant batch agree bitwise on
    # the shared problems?  No: the two layouts evaluate the same formulas from the same table values.
    mixed = np.concatenate([p[:8], sc["p"][:8]])
    a = bs.solve(mixed, np.tile([0.6, 0.1], (16, 20)))
    b = bs.solve(sc["p"][:8], np.tile([0.6, 0.1], (8, 20)))
    assert np.array_equal(a.solution[8:], b.solution)
    bs.close()


def test_edge_cases_and_error_behaviour(solver20, cfg20):
    sc = scenes.make_batch(cfg20, 3, n_dyn=1, seed=81)
    one = solver20.solve(sc["p"][0])                        # B = 1, 1-D input
    assert one.solution.shape == (1, 40)
    three = solver20.solve(sc["p"])
    assert np.array_equal(one.solution[0], three.solution[0])
    empty = solver20.solve(np.zeros((0, cfg20.num_params)))  # B = 0 is a no-op
    assert empty.solution.shape == (0, 40)
    with pytest.raises(MpcGpuError, match="3003"):
        solver20.solve(np.zeros((2, cfg20.num_params - 1)))
    with pytest.raises(MpcGpuError, match="1600"):
        solver20.solve(sc["p"], initial_guess=np.zeros((3, 39)))
    with pytest.raises(MpcGpuError, match="1700"):
        solver20.solve(sc["p"], initial_lagrange_multipliers=np.zeros((3, 41)))
    # non-finite parameters poison the iteration: reported as status 3 / None, never as a "solution"
    bad = sc["p"].copy(); bad[1, 0] = np.nan
    rb = solver20.solve(bad)
    assert rb.status[1] == 3 and rb.exit_status[1] == "NotFiniteComputation" and rb.status[0] != 3
    assert Solver(cfg20).run(bad[1].tolist(), None) is None
    # an all-zero parameter vector (every block padded) is still a valid problem
    z = solver20.solve(np.zeros((1, cfg20.num_params)))
    uo, _, ro, _ = oracle.solve_batch(oracle_cfg(cfg20), np.zeros((1, cfg20.num_params)))
    assert np.max(np.abs(z.solution - uo)) <= U_TOL and np.isfinite(z.cost[0])


def test_plugin_contract(cfg20):
    """``solver().run(p, initial_guess)`` -> .solution/.cost/.exit_status/.solve_time_ms
    (trajectory_generator.py:318-323)."""
    s = Solver(cfg20)
    sc = scenes.make_batch(cfg20, 1, n_dyn=0, with_box=False, seed=91)
    sol = s.run(sc["p"][0].tolist(), None)
    assert len(sol.solution) == 40 and isinstance(sol.cost, float)
    assert sol.exit_status in ("Converged", "NotConvergedIterations", "NotConvergedOutOfTime")
    assert sol.solve_time_ms > 0.0 and sol.num_outer_iterations >= 1
    assert s.run(sc["p"][0][:-1].tolist(), None) is None     # OpEn binding: wrong length -> None
    assert s.run(sc["p"][0].tolist(), [0.0] * 3) is None


def test_drop_in_module_is_importable_the_way_the_reference_loads_it(cfg20):
    """trajectory_generator.py:63-71: sys.path.append(<build_directory>/<optimizer_name>); __import__(name).solver()"""
    import importlib, os, sys
    from conftest import ROOT
    path = os.path.join(ROOT, cfg20.build_directory, cfg20.optimizer_name)
    sys.path.append(path)
    try:
        built_solver = importlib.import_module(cfg20.optimizer_name)
        s = built_solver.solver()
        sc = scenes.make_batch(cfg20, 1, n_dyn=1, with_box=False, seed=92, v_init_range=(1.0, 1.2))
        sol = s.run(sc["p"][0].tolist(), None)
        assert len(sol.solution) == 2 * cfg20.N_hor and sol.exit_status in ("Converged", "NotConvergedIterations")
    finally:
        sys.path.remove(path)


def test_device_pointer_entry_point(solver20, cfg20):
    import torch
    sc = scenes.make_batch(cfg20, 64, n_dyn=4, seed=95)
    host = solver20.solve(sc["p"])
    dev = torch.device("cuda:0")
    p = torch.from_numpy(sc["p"]).to(dev)
    out = dict(u=torch.empty(64, 40, dtype=torch.float64, device=dev),
               cost=torch.empty(64, dtype=torch.float64, device=dev),
               status=torch.empty(64, dtype=torch.int32, device=dev),
               inner_it=torch.empty(64, dtype=torch.int32, device=dev))
    solver20.solve_device(p, out, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(out["u"].cpu().numpy(), host.solution)
    assert np.array_equal(out["inner_it"].cpu().numpy(), host.num_inner_iterations)
    t = solver20.last_timing()
    assert t["solve_ms"] > 0.0


def test_closed_loop_batched_tracker_matches_single_robot_tracker_and_makes_progress(cfg20):
    """f1: B robots through BatchedTracker (one solve per tick) == B single-robot InterfaceMpc instances
    (bitwise: a robot's solve does not depend on its batch), and the closed loop behaves: robots advance along
    their path and stay out of the inflated box."""
    from trajtrack_mpcndqn_rlboost_amd import BatchedTracker, InterfaceMpc
    B, T = 6, 8
    box = [(6.7, 2.2), (9.3, 2.2), (9.3, 4.8), (6.7, 4.8)]           # scene-1 box inflated by 0.8 m
    paths = [[(0.6 + 0.1 * i, 3.5 + 0.2 * i), (15.4, 3.5 + 0.2 * i)] for i in range(B)]
    inits = [np.array([0.6 + 0.1 * i, 3.5 + 0.2 * i, 0.0]) for i in range(B)]
    goal = lambda i: np.array([15.4, 3.5 + 0.2 * i, 0.0])
    bt = BatchedTracker(cfg20, B)
    singles = []
    for i in range(B):
        bt.initialization(i, inits[i], goal(i), paths[i], "work")
        bt.update_static_constraints(i, [box])
        m = InterfaceMpc(cfg20, solver=Solver(cfg20))
        m.initialization(inits[i].copy(), goal(i), paths[i], "work")
        m.update_static_constraints([box])
        singles.append(m)
    for t in range(T):
        actions, pred, cost = bt.step("work")
        for i, m in enumerate(singles):
            ref, _ = m.get_local_ref_traj()
            a, p_, c_ = m.get_action(ref, mode="work")
            assert np.array_equal(a, actions[i]), (t, i)
            assert np.array_equal(np.array(p_), pred[i])
            assert np.array_equal(m.state, bt.states[i])
    assert np.all(bt.states[:, 0] > np.array([s[0] for s in inits]) + 0.5)       # moved forward
    for i in range(B):                                                          # predictions avoid the box interior
        inside = (pred[i][:, 0] > 6.7) & (pred[i][:, 0] < 9.3) & (pred[i][:, 1] > 2.2) & (pred[i][:, 1] < 4.8)
        assert inside.sum() <= 2


@pytest.mark.parametrize("N,n_dyn,B", [(20, 8, 8192), (40, 8, 4096), (20, 4, 1024)])
def test_full_size_configurations_through_size_independent_properties(N, n_dyn, B):
    """BASELINE.json configurations at their full batch sizes.  The oracle cannot solve thousands of problems in
    seconds, so the whole batch is checked through properties that do not need it: every control sequence lies in
    the input box; the reported cost is f(u) (oracle cost function, sampled); a re-solve of a random sub-batch in a
    different order is bitwise identical (batch-composition / sharding invariance); statuses and iteration counts
    are consistent with the caps; and a 48-problem sample agrees with the oracle like the small-batch tests."""
    cfg = make_cfg(N)
    ocfg = oracle_cfg(cfg)
    bs = BatchSolver(cfg)
    sc = scenes.make_batch(cfg, B, n_dyn=n_dyn, seed=1234)
    res = bs.solve(sc["p"])
    uu = res.solution.reshape(B, N, 2)
    assert np.all(np.isfinite(res.solution)) and np.all(np.isfinite(res.cost))
    assert uu[..., 0].min() >= cfg.lin_vel_min and uu[..., 0].max() <= cfg.lin_vel_max
    assert np.abs(uu[..., 1]).max() <= cfg.ang_vel_max
    assert set(np.unique(res.status)) <= {0, 1}
    assert res.num_outer_iterations.min() >= 1 and res.num_outer_iterations.max() <= cfg.solver_max_outer_iterations
    assert res.num_inner_iterations.max() <= cfg.solver_max_outer_iterations * cfg.solver_max_inner_iterations
    assert np.all(res.num_outer_iterations[res.status == 0] < cfg.solver_max_outer_iterations)
    rng = np.random.default_rng(B)
    pick = rng.choice(B, 48, replace=False)
    for i in pick[:16]:
        assert _rel(res.cost[i], oracle.cost_grad(ocfg, res.solution[i], sc["p"][i])["f"]) < 1e-10
    again = bs.solve(sc["p"][pick[::-1]])
    assert np.array_equal(again.solution, res.solution[pick[::-1]])
    assert np.array_equal(again.num_inner_iterations, res.num_inner_iterations[pick[::-1]])
    uo, _, ro, _ = oracle.solve_batch(ocfg, sc["p"][pick])
    assert (res.status[pick] == ro["status"]).mean() >= 0.85
    assert abs(res.num_inner_iterations[pick].mean() / ro["inner_iters"].mean() - 1.0) < 0.15
    # capped solves land in different local basins (costs span an order of magnitude): the per-problem cost ratio
    # must scatter around 1 without bias
    assert abs(np.median(np.log(res.cost[pick] / ro["cost"]))) < 0.25
    both = (res.status[pick] == 0) & (ro["status"] == 0)
    if both.any():
        assert np.max(np.abs(res.solution[pick] - uo), axis=1)[both].max() <= U_TOL
    bs.close()


@pytest.mark.parametrize("N", [2, 5, 16, 17, 21, 22, 32, 33, 48, 64])
def test_generic_horizons_cover_every_lane_mapping(N):
    """Horizons without a compiled specialisation run the generic kernel; the lanes-per-step split changes at
    N = 16 / 21 / 32 and the DPP row count with it.  Cost/gradient against the oracle plus a short tracked solve."""
    cfg = make_cfg(N)
    ocfg = oracle_cfg(cfg)
    bs = BatchSolver(cfg)
    B = 12
    n_dyn = min(6, cfg.Ndynobs)
    if N >= 8:
        sc = scenes.make_batch(cfg, B, n_dyn=n_dyn, n_other=2, seed=300 + N)
        p = sc["p"]
    else:  # scenes need a few steps of horizon for their geometry; tiny horizons get random but valid parameters
        rng0 = np.random.default_rng(N)
        p = np.zeros((B, cfg.num_params))
        off = cfg.offsets()
        p[:, 0:3] = rng0.uniform(-1, 1, (B, 3)); p[:, 8:18] = [0, 10, 0, 0.5, 0.5, 2, 1, 100, 10, 20]
        ref = np.cumsum(np.full((B, N, 1), 0.24), axis=1) * np.array([1.0, 0.2, 0.0]) + p[:, None, 0:3] * [1, 1, 0]
        p[:, off["r"]:off["r"] + 3 * N] = ref.reshape(B, 3 * N); p[:, 3:6] = ref[:, -1]
        p[:, off["vref"]:off["vref"] + N] = 1.0
        p[:, off["od"]:off["od"] + 6 * N] = np.tile([0.3, 0.1, 0.8, 0.5, 0.2, 1.0], N)
        p[:, off["qdyn"]:off["qdyn"] + N] = 1e3
    rng = np.random.default_rng(N)
    u = np.stack([rng.uniform(-0.6, 1.6, (B, N)), rng.uniform(-0.6, 0.6, (B, N))], axis=2).reshape(B, 2 * N)
    c = rng.choice([0.0, 10.0, 250.0], B)
    y = rng.uniform(-2, 2, (B, 2 * N))
    r = bs.cost_grad(u, p, c, y)
    for i in range(B):
        o = oracle.cost_grad(ocfg, u[i], p[i], float(c[i]), y[i])
        assert _rel(r["psi"][i], o["psi"]) < RTOL_COST and _rel(r["f"][i], o["f"]) < RTOL_COST
        assert _rel(r["grad"][i], o["grad"]) < RTOL_COST
        assert _rel(r["F1"][i], o["F1"]) < RTOL_COST and _rel(r["F2"][i], o["F2"]) < RTOL_COST
    bs.close()
    cfgk = make_cfg(N, solver_max_inner_iterations=5, solver_max_outer_iterations=2)
    bs = BatchSolver(cfgk)
    u0 = np.tile([0.6, 0.1], (B, N))
    res = bs.solve(p, u0)
    uo, _, ro, _ = oracle.solve_batch(oracle_cfg(cfgk), p, u0)
    assert np.array_equal(res.num_inner_iterations, ro["inner_iters"])
    du = np.max(np.abs(res.solution - uo), axis=1)
    assert np.median(du) < 1e-7 and du.max() < 1e-3      # one stiff problem may already have amplified rounding
    bs.close()


def test_large_lds_carve_long_horizon_with_time_varying_obstacles():
    """N = 64 with 15 time-varying ellipses needs > 64 KiB of LDS per wavefront (general 9-doubles-per-item tables):
    the library opts the kernels into the larger dynamic-LDS limit.  An impossible carve is reported, not launched."""
    N = 64
    cfg = make_cfg(N)
    ocfg = oracle_cfg(cfg)
    off = cfg.offsets()
    B = 6
    sc = scenes.make_batch(cfg, B, n_dyn=15, n_other=3, seed=77)
    p = sc["p"].copy()
    od = p[:, off["od"]:off["od"] + 15 * 6 * N].reshape(B, 15, N, 6)
    od[..., 2] = 0.5 + 0.01 * np.arange(N)
    od[..., 4] = 0.02 * np.arange(N)
    bs = BatchSolver(cfg)
    rng = np.random.default_rng(1)
    u = np.stack([rng.uniform(-0.5, 1.5, (B, N)), rng.uniform(-0.5, 0.5, (B, N))], axis=2).reshape(B, 2 * N)
    r = bs.cost_grad(u, p, np.full(B, 10.0))
    assert bs.last_shape()["lds_bytes"] > 64 * 1024
    for i in range(B):
        o = oracle.cost_grad(ocfg, u[i], p[i], 10.0)
        assert _rel(r["psi"][i], o["psi"]) < RTOL_COST and _rel(r["grad"][i], o["grad"]) < RTOL_COST
    bs.close()
    cfgk = make_cfg(N, solver_max_inner_iterations=4, solver_max_outer_iterations=1)
    bs = BatchSolver(cfgk)
    u0 = np.tile([0.6, 0.1], (B, N))
    res = bs.solve(p, u0)
    uo, _, ro, _ = oracle.solve_batch(oracle_cfg(cfgk), p, u0)
    assert np.array_equal(res.num_inner_iterations, ro["inner_iters"]) and np.max(np.abs(res.solution - uo)) < 1e-6
    bs.close()
    big = make_cfg(N, Ndynobs=32, Nother=16)
    bs = BatchSolver(big)
    pb = np.zeros((1, big.num_params))
    ob = big.offsets()
    blk = np.ones((32, N, 6)) * np.arange(1, N + 1)[None, :, None]      # every row active and time-varying
    pb[0, ob["od"]:ob["od"] + 32 * 6 * N] = blk.reshape(-1)
    pb[0, ob["c"]:ob["os"]] = 1.0                                        # ... and every other-robot row too (round 3: the hinge
    #                                                                      matrix left LDS, 32 general rows alone fit again)
    with pytest.raises(MpcGpuError, match="LDS carve"):
        bs.solve(pb)
    bs.close()


def test_both_register_allocation_variants_give_bitwise_identical_solutions():
    """The launcher runs the 148-VGPR build (3 wavefronts/SIMD) for batches that fit 12 problems per CU or whose LDS
    carve exceeds 10 KiB, and the 128-VGPR build (4 wavefronts/SIMD, a few spilled registers) otherwise.  Same
    arithmetic: the same problems must come out bitwise equal from both."""
    cfg = make_cfg(20)
    bs = BatchSolver(cfg, latency_batch=0)          # the throughput kernel's two builds (small batches would take the latency kernel)
    big = scenes.make_batch(cfg, 4096, n_dyn=8, seed=77)
    res_big = bs.solve(big["p"])
    shape = bs.last_shape()
    assert shape["lds_bytes"] <= 10 * 1024 and shape["waves_per_simd"] == 4
    small = bs.solve(big["p"][:256])
    assert bs.last_shape()["waves_per_simd"] == 3
    assert np.array_equal(small.solution, res_big.solution[:256])
    assert np.array_equal(small.cost, res_big.cost[:256])
    assert np.array_equal(small.num_inner_iterations, res_big.num_inner_iterations[:256])
    # many time-varying obstacles: the carve is beyond 10 KiB, the batch size no longer matters
    dense = scenes.make_batch(cfg, 4096, n_dyn=15, seed=78)
    bs.solve(dense["p"])
    assert bs.last_shape()["lds_bytes"] > 10 * 1024 and bs.last_shape()["waves_per_simd"] == 3


def test_fleet_coupling_keeps_crossing_robots_apart():
    """scenario_simulator.py semantics in batch: groups of 2 robots whose straight paths cross at the same time.  With
    the other robot's previous prediction in the parameter vector (fleet term, mpc_generator.py:211-216) they pass each
    other at more than the vehicle width; without it they run through each other."""
    from trajtrack_mpcndqn_rlboost_amd import BatchedTracker
    cfg = make_cfg(20)
    G = 8                                               # 8 independent worlds of 2 robots each
    def run(share):
        bt = BatchedTracker(cfg, 2 * G)
        for g in range(G):
            y = 3.0 + 0.5 * g
            bt.initialization(2 * g, np.array([0.0, y, 0.0]), np.array([8.0, y, 0.0]), [(0.0, y), (8.0, y)], "work")
            bt.initialization(2 * g + 1, np.array([8.0, y + 0.05, np.pi]), np.array([0.0, y + 0.05, 0.0]),
                              [(8.0, y + 0.05), (0.0, y + 0.05)], "work")
        dmin = np.full(G, np.inf)
        for t in range(45):
            if share:
                bt.share_predictions([[2 * g, 2 * g + 1] for g in range(G)])
            bt.step("work")
            d = np.hypot(*(bt.states[0::2, :2] - bt.states[1::2, :2]).T)
            dmin = np.minimum(dmin, d)
        return dmin, bt.states.copy()
    apart, st = run(True)
    through, _ = run(False)
    print("min distance with / without sharing:", np.round(apart, 3), np.round(through, 3))
    assert through.max() < 0.25                          # head-on along (almost) the same line
    # the fleet term is a soft cost on the PREVIOUS tick's prediction of the other robot (Jacobi), so the clearance is
    # not guaranteed per pair; it acts below vehicle_width (0.5 m) and most pairs settle right at it
    assert np.median(apart) > cfg.vehicle_width * 0.9 and (apart > through + 0.1).all()
    assert (st[0::2, 0] > 5.0).all() and (st[1::2, 0] < 3.0).all()   # and everybody got past
