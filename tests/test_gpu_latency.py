"""GPU (-m gpu): the latency kernel (csrc/mpc_team.hpp: one problem per workgroup of four wavefronts, Lipschitz test and
line-search trials of a PANOC step evaluated side by side, compaction fused, no count read-back) against the throughput
kernel (one problem per wavefront, sequential evaluations).  Same device functions on the same inputs: every output must be
BITWISE equal -- solutions, costs, statuses, iteration counts, multipliers and the evaluation counts of the sequential
algorithm.  The reference's own call pattern is one problem per call (src/interface_mpc.py:82-88)."""
import numpy as np
import pytest

from conftest import make_cfg
from trajtrack_mpcndqn_rlboost_amd import BatchSolver, scenes

pytestmark = pytest.mark.gpu


def _same(a, b):
    assert np.array_equal(a.solution, b.solution)
    assert np.array_equal(a.cost, b.cost, equal_nan=True)
    assert np.array_equal(a.status, b.status)
    assert np.array_equal(a.num_inner_iterations, b.num_inner_iterations)
    assert np.array_equal(a.num_outer_iterations, b.num_outer_iterations)
    assert np.array_equal(a.last_problem_norm_fpr, b.last_problem_norm_fpr)
    assert np.array_equal(a.f2_norm, b.f2_norm)
    assert np.array_equal(a.lagrange_multipliers, b.lagrange_multipliers)


FAMILIES = {
    "benchmark (cap-limited, hard constraints active)": dict(n_dyn=8),
    "passing (about half converge)": dict(n_dyn=8, dyn_clearance=0.1, box_clearance=0.3),
    "free corridor (converge quickly)": dict(n_dyn=0, with_box=False, v_init_range=(1.0, 1.2)),
    "fleet + few obstacles": dict(n_dyn=3, n_other=2),
}


@pytest.mark.parametrize("family", list(FAMILIES))
@pytest.mark.parametrize("fallback,stall", [("last_trial", "either"), ("half_step", "either"), ("last_trial", "both"), ("half_step", "both")])
def test_latency_kernel_is_bitwise_equal_to_the_throughput_kernel(family, fallback, stall):
    cfg = make_cfg(20, solver_linesearch_fallback=fallback, solver_penalty_stall=stall)
    B = 40
    sc = scenes.make_batch(cfg, B, seed=97, **FAMILIES[family])
    fast = BatchSolver(cfg)                       # library rule: latency kernel for small batches
    seq = BatchSolver(cfg, latency_batch=0)       # throughput kernel only
    for u0 in (None, np.tile([0.6, 0.1], (B, 20))):
        a, b = fast.solve(sc["p"], u0), seq.solve(sc["p"], u0)
        assert fast.last_shape()["latency_kernel"] and not seq.last_shape()["latency_kernel"]
        _same(a, b)
        ea, eb = fast.last_eval_counts(B), seq.last_eval_counts(B)
        assert np.array_equal(ea[0], eb[0]) and np.array_equal(ea[1], eb[1])
    print(f"\n[{family}, {fallback}, {stall}] statuses {np.bincount(a.status, minlength=3).tolist()}, inner iterations "
          f"{a.num_inner_iterations.min()}..{a.num_inner_iterations.max()}")
    fast.close(); seq.close()


def test_latency_kernel_with_short_caps_warm_start_and_multipliers():
    """Iteration caps that end inner problems mid-way, a warm start with multipliers and penalty handed in."""
    cfg = make_cfg(20, solver_max_inner_iterations=7, solver_max_outer_iterations=3)
    B = 24
    sc = scenes.make_batch(cfg, B, n_dyn=6, seed=5)
    rng = np.random.default_rng(5)
    u0 = np.stack([rng.uniform(0.0, 1.2, (B, 20)), rng.uniform(-0.4, 0.4, (B, 20))], axis=2).reshape(B, 40)
    y0 = rng.normal(0.0, 0.3, (B, 40))
    c0 = rng.uniform(5.0, 200.0, B)
    fast, seq = BatchSolver(cfg), BatchSolver(cfg, latency_batch=0)
    _same(fast.solve(sc["p"], u0, y0, c0), seq.solve(sc["p"], u0, y0, c0))
    assert fast.last_shape()["latency_kernel"]
    fast.close(); seq.close()


@pytest.mark.parametrize("waves", [4, 2])
def test_inner_problems_without_a_step(waves):
    """Round 5.  With an inner cap near the length of the first inner problem, many second inner problems meet their tolerance at
    the first residual and take NO PANOC step.  A build of the latency kernel whose loop-carried scalars had been packed by the
    SLP vectoriser lost the iteration total and moved the L-BFGS ring position exactly there (and parted from the throughput
    kernel by an ulp in the next inner problem): csrc/Makefile, -fno-slp-vectorize.  Also: the avoidance family at full caps."""
    n = 1000
    base = make_cfg(20)
    p = scenes.make_family(base, n, "avoidance", n_dyn=8, seed=320)["p"]
    zero_step = 0
    for caps in ((3, 139), (3, 140), (4, 60), (10, 500)):
        cfg = make_cfg(20, solver_max_outer_iterations=caps[0], solver_max_inner_iterations=caps[1])
        seq = BatchSolver(cfg, latency_batch=0, tail_promotion=0)
        fast = BatchSolver(cfg, latency_batch=n, tail_promotion=0)
        fast.set_tail_promotion(0, waves=waves)
        for bs in (seq, fast):
            bs.reserve_shape(max_static=5, max_fleet=0, max_dyn=8, var_shape=True)
        a, b = seq.solve(p), fast.solve(p)
        assert int(fast._L.mpcgpu_last_latency_kernel(fast._h)) == waves and not seq.last_shape()["latency_kernel"]
        _same(a, b)
        ea, eb = seq.last_eval_counts(n), fast.last_eval_counts(n)
        assert np.array_equal(ea[0], eb[0]) and np.array_equal(ea[1], eb[1])
        if caps[0] == 3:   # problems whose second inner problem took no step: 3 evaluations (initial point, Lipschitz estimate, outer step)
            two = BatchSolver(make_cfg(20, solver_max_outer_iterations=2, solver_max_inner_iterations=caps[1]), latency_batch=0, tail_promotion=0)
            one = BatchSolver(make_cfg(20, solver_max_outer_iterations=1, solver_max_inner_iterations=caps[1]), latency_batch=0, tail_promotion=0)
            r2, r1 = two.solve(p), one.solve(p)
            zero_step += int(np.sum((r2.num_outer_iterations == 2) & (r2.num_inner_iterations == r1.num_inner_iterations)))
            two.close(); one.close()
        seq.close(); fast.close()
    assert zero_step >= 5, zero_step


def test_selection_rule_and_single_problem_call():
    cfg = make_cfg(20)
    sc = scenes.make_batch(cfg, 2048, n_dyn=4, seed=3, dyn_clearance=0.1, box_clearance=0.3)
    bs = BatchSolver(cfg)
    big = bs.solve(sc["p"])
    assert not bs.last_shape()["latency_kernel"]                  # 2048 > 2 x 256 compute units
    one = bs.solve(sc["p"][17])
    assert bs.last_shape()["latency_kernel"]
    assert np.array_equal(one.solution[0], big.solution[17]) and one.status[0] == big.status[17]
    few = bs.solve(sc["p"][100:108])
    assert np.array_equal(few.solution, big.solution[100:108])
    bs.close()


@pytest.mark.parametrize("N,caps", [(40, (500, 10)), (12, (60, 4)), (33, (40, 3))])
def test_latency_kernel_for_other_horizons(N, caps):
    """Compiled horizon 40 and the runtime-horizon instantiation (12, 33): bitwise equal to the throughput kernel too."""
    cfg = make_cfg(N, solver_max_inner_iterations=caps[0], solver_max_outer_iterations=caps[1])
    B = 12
    sc = scenes.make_batch(cfg, B, n_dyn=5, seed=40 + N, dyn_clearance=0.1, box_clearance=0.3)
    fast, seq = BatchSolver(cfg), BatchSolver(cfg, latency_batch=0)
    for u0 in (None, np.tile([0.6, 0.1], (B, N))):
        a, b = fast.solve(sc["p"], u0), seq.solve(sc["p"], u0)
        assert fast.last_shape()["latency_kernel"] and not seq.last_shape()["latency_kernel"]
        _same(a, b)
    fast.close(); seq.close()


@pytest.mark.parametrize("kernel", ["latency", "throughput", "two-per-wavefront"])
def test_wall_clock_limit_ends_the_solve_with_its_own_status(kernel):
    """`with_max_duration_micros` (mpc_generator.py:22,287; OpEn: NotConvergedOutOfTime).  The in-kernel limit reads the
    100 MHz wall clock; in the latency kernel wavefront 0 decides for the whole workgroup (the replicas must never disagree,
    or the workgroup would hang at a barrier)."""
    cfg = make_cfg(20, solver_max_duration_micros=3000)          # 3 ms: far below what a cap-limited solve needs
    B = 24
    hard = scenes.make_batch(cfg, B, n_dyn=8, seed=11)["p"]
    kw = dict(latency_batch=0) if kernel == "throughput" else (dict(pairing=2) if kernel == "two-per-wavefront" else {})
    bs = BatchSolver(cfg, **kw)
    res = bs.solve(hard)
    assert bs.last_shape()["latency_kernel"] == (kernel == "latency")
    # (under the "either" reading of the stall rule a few of these keep c = 10, solve ten easy inner problems inside the 3 ms and end
    #  at the OUTER cap -- status 1 before the clock -- instead)
    in_time = (res.status == 1) & (res.solve_time_ms < 3.0)
    assert np.all((res.status == 2) | in_time), np.bincount(res.status, minlength=5)
    assert np.mean(res.status == 2) >= 0.7
    assert bs.last_timing()["solve_ms"] < 50.0                   # it did stop (a full solve of these takes 40-140 ms)
    assert np.all(np.isfinite(res.solution)) and np.all(np.isfinite(res.cost))
    u = res.solution.reshape(B, 20, 2)
    assert u[..., 0].min() >= cfg.lin_vel_min - 1e-12 and u[..., 0].max() <= cfg.lin_vel_max + 1e-12   # the feasible half step
    assert np.abs(u[..., 1]).max() <= cfg.ang_vel_max + 1e-12
    assert np.all(res.solve_time_ms < 50.0) and np.all(res.solve_time_ms > 1.0)
    bs.close()
    # no limit: the same problems run to their iteration caps
    free = BatchSolver(make_cfg(20, solver_max_duration_micros=0, solver_max_inner_iterations=30, solver_max_outer_iterations=2), **kw)
    assert np.all(free.solve(hard).status == 1)
    free.close()


@pytest.mark.parametrize("N,B,caps", [(20, 700, (120, 4)), (20, 1024, (60, 3)), (40, 600, (60, 3))])
def test_mid_batches_run_two_wavefronts_per_problem_bitwise_equal_too(N, B, caps):
    """Between two and four problems per compute unit (round 3): the compaction is its own launch, the tables are sized from the
    batch's maxima and a problem gets TWO wavefronts (Lipschitz test + tau = 1 side by side, then two trials per pass), so the
    whole batch is resident at once.  Same device functions: every output is bitwise that of the throughput kernel."""
    cfg = make_cfg(N, solver_max_inner_iterations=caps[0], solver_max_outer_iterations=caps[1])
    sc = scenes.make_batch(cfg, B, n_dyn=4, seed=123, dyn_clearance=0.1, box_clearance=0.3)
    fast, seq = BatchSolver(cfg, latency_batch=1024), BatchSolver(cfg, latency_batch=0)     # (MPCGPU_OPT_TEAM_BATCH = 1024: the mid-range form)
    a, b = fast.solve(sc["p"]), seq.solve(sc["p"])
    took = fast._L.mpcgpu_last_latency_kernel(fast._h)
    # N_hor = 20: four two-wavefront workgroups fit a compute unit; N_hor = 40: they do not, the throughput kernel runs
    assert took == (2 if N == 20 else 0) and seq._L.mpcgpu_last_latency_kernel(seq._h) == 0
    assert fast.last_shape()["max_dyn"] == 4                      # tables from the batch, not from the configured maxima
    _same(a, b)
    ea, eb = fast.last_eval_counts(B), seq.last_eval_counts(B)
    assert np.array_equal(ea[0], eb[0]) and np.array_equal(ea[1], eb[1])
    fast.close(); seq.close()


def test_reserved_small_batch_takes_the_latency_kernel_with_tables_of_the_reservation():
    import torch
    cfg = make_cfg(20, solver_max_inner_iterations=80, solver_max_outer_iterations=3)
    B = 200
    sc = scenes.make_batch(cfg, B, n_dyn=3, seed=9, dyn_clearance=0.1, box_clearance=0.3)
    dev = torch.device("cuda", 0)
    p = torch.from_numpy(sc["p"]).to(dev)
    out = dict(u=torch.empty(B, 40, dtype=torch.float64, device=dev), cost=torch.empty(B, dtype=torch.float64, device=dev),
               status=torch.empty(B, dtype=torch.int32, device=dev))
    bs = BatchSolver(cfg)
    bs.reserve_shape(max_static=5, max_fleet=0, max_dyn=2, var_shape=True)        # one row too few for these scenes
    bs.solve_device(p, out, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert bs._L.mpcgpu_last_latency_kernel(bs._h) == 4
    assert (out["status"] == 4).all()                                             # reported, not solved
    bs.reserve_shape(max_static=5, max_fleet=0, max_dyn=3, var_shape=True)
    bs.solve_device(p, out, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    ref = BatchSolver(cfg, latency_batch=0).solve(sc["p"])
    assert np.array_equal(out["u"].cpu().numpy(), ref.solution) and np.array_equal(out["status"].cpu().numpy(), ref.status)
    bs.close()


@pytest.mark.parametrize("B", [40, 700])
def test_latency_kernel_decision_trace_is_bitwise_the_throughput_kernel_trace(B):
    """Trace builds record one line per PANOC step (penalty, Lipschitz estimate, step size, residual, psi, doublings, pairs,
    halvings, tau): the four- and two-wavefront kernels write the very lines the one-wavefront kernel writes."""
    from trajtrack_mpcndqn_rlboost_amd.solver import variant_path
    cfg = make_cfg(20, solver_max_inner_iterations=40, solver_max_outer_iterations=4)
    sc = scenes.make_batch(cfg, B, n_dyn=8, seed=55)
    out = []
    for kw in (dict(latency_batch=1024), dict(latency_batch=0)):
        bs = BatchSolver(cfg, library=variant_path("trace"), **kw)
        bs.set_trace(160)
        res = bs.solve(sc["p"], np.tile([0.6, 0.1], (B, 20)))
        out.append((res, bs.read_trace(B), bs._L.mpcgpu_last_latency_kernel(bs._h)))
        bs.close()
    assert out[0][2] == (4 if B <= 512 else 2) and out[1][2] == 0
    _same(out[0][0], out[1][0])
    assert np.array_equal(out[0][1], out[1][1], equal_nan=True)
    assert np.isfinite(out[0][1][:, :20]).all() and np.isfinite(out[0][1][:, 100]).mean() > 0.5   # the steps were recorded
