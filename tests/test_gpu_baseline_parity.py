"""GPU (-m gpu): the north-star tolerance asserted on the BASELINE.json configurations themselves, and decision-level
parity of the solver iteration against the oracle.

* `test_baseline_configurations_converged_solves_match_oracle`: configs 2, 3 and the metric configuration at their full
  batch size on the "passing" scene family (same N_hor, same active-row counts, same batch; discs and box stand beside
  the path so a collision-free plan exists).  A HARD MINIMUM of problems must converge on both sides and every one of
  them must agree with the oracle within |du|_inf <= 1e-3 (north_star's tolerance).
* `test_decision_trace_matches_oracle`: a -DMPC_TRACE build of the library writes one record per PANOC step (outer index,
  c, L, gamma, ||gamma fpr||, psi, Lipschitz doublings, L-BFGS pairs, line-search halvings, tau); the oracle emits the
  same trace.  On obstacle scenes from a non-zero initial guess with several outer iterations every DISCRETE decision
  must be identical up to the first divergence index (reported), the scalars must agree before it, and the first
  divergence must come late.
* both line-search fallback readings, the L-BFGS-in-LDS build (north_star's layout) bitwise equal
  to the product build, stream ordering of the device-pointer entry point, reserved LDS carves.
"""
import numpy as np
import pytest

import oracle
from conftest import make_cfg, oracle_cfg
from trajtrack_mpcndqn_rlboost_amd import BatchSolver, scenes
from trajtrack_mpcndqn_rlboost_amd.solver import variant_path

pytestmark = pytest.mark.gpu
U_TOL = 1e-3  # north_star tolerance on control sequences

DISCRETE = [0, 1, 7, 8, 9]          # outer, step, Lipschitz doublings, L-BFGS pairs, halvings
SCALARS = [2, 3, 4, 5, 6, 10, 11]   # c, L, gamma, ||gamma fpr||, psi(u), tau, psi(u+)
SCALAR_TOL = 1e-3


# (N_hor, n_dyn, B, minimum fraction of the oracle sample that must converge on BOTH sides, minimum agreement on which do).
# N = 40: the AKKT test ||gamma fpr|| / gamma < eps is rarely met within 500 inner iterations with 80 unknowns (free space
# alone: 30 %; this family: 9 %), and WHETHER it is met is itself at the mercy of rounding: the oracle against ITSELF with
# every parameter moved by one ulp agrees on 5 of its 23 converged problems (measured).  The status-based floor is
# therefore small there, and the tolerance is asserted on a second, much larger set: the problems whose OUTER loop ended
# by its own criteria on both sides (constraints and multipliers within tolerance; at most the last inner problem ran into
# its iteration cap, with ||gamma fpr|| < eps).
@pytest.mark.parametrize("N,n_dyn,B,min_frac,min_agree", [(20, 8, 8192, 0.25, 0.9), (40, 8, 4096, 0.01, 0.8),
                                                           (20, 4, 1024, 0.25, 0.9)])
def test_baseline_configurations_converged_solves_match_oracle(N, n_dyn, B, min_frac, min_agree):
    cfg = make_cfg(N)
    ocfg = oracle_cfg(cfg)
    bs = BatchSolver(cfg)
    sc = scenes.make_batch(cfg, B, n_dyn=n_dyn, seed=4321, dyn_clearance=0.1, box_clearance=0.3)
    res = bs.solve(sc["p"])
    shape = bs.last_shape()
    # the latency kernel takes batches up to two problems per compute unit (512); config 2's batch (1024) runs the throughput kernel, which
    # promotes all of it (round 6: faster than the two-wavefront mid-range form under the default stall reading, profiles/r06_team_sweep.txt)
    assert bool(shape["latency_kernel"]) == (B <= 512)
    assert shape["max_dyn"] == n_dyn and shape["max_static"] == 5          # same active rows as the benchmark family
    S = 256
    pick = np.random.default_rng(N + n_dyn).choice(B, S, replace=False)
    uo, _, ro, _ = oracle.solve_batch(ocfg, sc["p"][pick])
    both = (res.status[pick] == 0) & (ro["status"] == 0)
    du = np.max(np.abs(res.solution[pick] - uo), axis=1)
    cap = cfg.solver_max_outer_iterations
    ended = (res.num_outer_iterations[pick] < cap) & (ro["outer_iters"] < cap) & \
            (res.last_problem_norm_fpr[pick] < cfg.solver_tolerance) & (ro["fpr"] < cfg.solver_tolerance)
    print(f"\n[N={N} n_dyn={n_dyn} B={B}] converged GPU {np.mean(res.status == 0):.3f} (whole batch) / "
          f"{np.mean(res.status[pick] == 0):.3f} (sample), oracle {np.mean(ro['status'] == 0):.3f}; on both {both.sum()}/{S}: "
          f"|du|inf max {du[both].max() if both.any() else float('nan'):.2e}, "
          f"median {np.median(du[both]) if both.any() else float('nan'):.2e}; outer loop ended by its criteria on both "
          f"{ended.sum()}/{S}: |du|inf median {np.median(du[ended]):.2e}, p90 {np.quantile(du[ended], 0.9):.2e}, max {du[ended].max():.2e}")
    assert both.sum() >= max(3, min_frac * S), (both.sum(), S)
    assert du[both].max() <= U_TOL
    # "both" reading: the penalty grows until ||F2|| <= delta, 80 % of the outer loops end by their own criteria; "either" (the default):
    # a problem whose plan touches a hard ellipse keeps c = 10 and runs into the outer cap -- the set shrinks to about the converged ones
    assert ended.sum() >= (0.8 if cfg.solver_penalty_stall == "both" else 0.3 if N == 20 else 0.02) * S
    assert np.quantile(du[ended], 0.9) <= U_TOL and np.median(du[ended]) <= 1e-4
    assert du[ended].max() <= 1e-2          # the oracle's own 1-ulp sensitivity on this set is 1.3e-3 .. 2.1e-3 (measured)
    # the two sides also agree on WHICH problems converge (those at the edge of an iteration cap flip, see above)
    assert np.mean((res.status[pick] == 0) == (ro["status"] == 0)) >= min_agree
    # outer-iteration paths of the converged problems are the same decisions
    assert np.array_equal(res.num_outer_iterations[pick][both], ro["outer_iters"][both])
    assert abs(np.mean(res.status == 0) - np.mean(res.status[pick] == 0)) < 0.1
    bs.close()


# Round 4.  (a) The "avoidance" family (scenes.FAMILIES): 1-3 discs COVER the reference path, the box covers it in 30 % of the
# problems, the detour leads to the free side of the corridor -- the problems this MPC exists for.  (b) Config 3 with a robust
# set of converged problems: at N_hor = 40 WHICH problems meet the AKKT test within 500 inner iterations is decided by rounding
# (the oracle against itself with every parameter moved by one ulp: 52 and 52 of 256 converge on the "on_track" family, 8 in
# common), so the tolerance is asserted with the inner cap raised to 5000 (`solver_max_inner_iterations`, the optional key of
# the yaml surface; mean 2400 iterations are used) where 93 % converge on both sides; the default caps are covered by the test
# above.  Next to every GPU-vs-oracle figure the test prints the oracle-vs-1-ulp-perturbed-oracle figure on the same sample.
@pytest.mark.parametrize("N,family,B,cfgkw,min_both,min_agree", [
    (20, "avoidance", 8192, {}, 77, 0.9),
    (40, "on_track", 4096, dict(solver_max_inner_iterations=5000), 48, 0.9)])
def test_avoidance_and_long_iteration_families_match_oracle(N, family, B, cfgkw, min_both, min_agree):
    cfg = make_cfg(N, **cfgkw)
    ocfg = oracle_cfg(cfg)
    bs = BatchSolver(cfg)
    sc = scenes.make_family(cfg, B, family, seed=4321)
    res = bs.solve(sc["p"])
    assert bs.last_shape()["max_dyn"] == 8 and bs.last_shape()["max_static"] == 5
    S = 256
    pick = np.random.default_rng(N).choice(B, S, replace=False)
    uo, _, ro, _ = oracle.solve_batch(ocfg, sc["p"][pick])
    up, _, rp, _ = oracle.solve_batch(ocfg, np.nextafter(sc["p"][pick], np.inf))      # the oracle's own sensitivity
    both = (res.status[pick] == 0) & (ro["status"] == 0)
    both_p = (rp["status"] == 0) & (ro["status"] == 0)
    du = np.max(np.abs(res.solution[pick] - uo), axis=1)
    dp = np.max(np.abs(up - uo), axis=1)
    agree = np.mean((res.status[pick] == 0) == (ro["status"] == 0))
    agree_p = np.mean((rp["status"] == 0) == (ro["status"] == 0))
    print(f"\n[{family} N={N} B={B} {cfgkw}] status histogram (whole batch) {np.bincount(res.status, minlength=3).tolist()}; sample of {S}: "
          f"converged GPU {np.sum(res.status[pick] == 0)}, oracle {np.sum(ro['status'] == 0)}, on both {both.sum()}, |du|inf max "
          f"{du[both].max():.2e} median {np.median(du[both]):.2e}, agreement on which converge {agree:.3f}  ||  oracle vs oracle with "
          f"every parameter moved by one ulp: on both {both_p.sum()}, |du|inf max {dp[both_p].max():.2e} median "
          f"{np.median(dp[both_p]):.2e}, agreement {agree_p:.3f}")
    assert both.sum() >= min_both, (both.sum(), S)
    assert du[both].max() <= U_TOL
    assert agree >= min_agree
    assert np.array_equal(res.num_outer_iterations[pick][both], ro["outer_iters"][both])
    assert np.mean(res.status == 0) >= 0.30                       # >= 30 % of the cold-start solves converge
    bs.close()


def first_divergence(tg, to):
    """Index of the first PANOC step whose discrete decisions differ (min(len) when none does)."""
    n = min(len(tg), len(to))
    d = np.any(tg[:n][:, DISCRETE] != to[:n][:, DISCRETE], axis=1)
    return int(np.argmax(d)) if d.any() else n


# Thresholds (median first discrete divergence, median first scalar drift beyond 1e-3, worst relative scalar error over
# steps 0..9) are the measured behaviour with a margin.  With 80 unknowns (N = 40) rounding differences are amplified
# faster: decisions flip after ~40 steps instead of ~70 (the oracle against itself with 1-ulp perturbed inputs does the same).
@pytest.mark.parametrize("N,fallback,max_inner,max_outer,min_fd,min_sd,early_tol,kernel,cold", [
    (20, "last_trial", 40, 6, 50, 30, 1e-6, "throughput", False),   # 240 steps across 6 inner problems: c = 10 .. 10*5^5, multipliers updated 5 times
    (20, "half_step", 40, 6, 50, 30, 1e-6, "throughput", False),
    (40, "last_trial", 40, 6, 30, 20, 1e-5, "throughput", False),
    (20, "last_trial", 500, 10, 50, 30, 1e-6, "throughput", False),  # the yaml's caps: the first 240 steps of the first inner problem
    # round 3: the LATENCY kernel's own trace (four wavefronts per problem, speculative evaluations) against the oracle ...
    (20, "last_trial", 40, 6, 50, 30, 1e-6, "latency", False),
    (40, "last_trial", 40, 6, 30, 20, 1e-5, "latency", False),
    # ... and the reference's real call pattern: cold start u0 = 0 (src/interface_mpc.py:82) with the yaml's caps.  The Lipschitz
    # estimate then perturbs by h = 1e-12, so L -- and with it gamma and every scalar after it -- is rounding-noise limited
    # (~1e-4 relative) in ANY float64 implementation: the scalars part at once (no early_tol / min_sd), the DISCRETE decisions
    # still coincide for a median of 20 steps (measured; profiles/archive/r03_parity_report.txt)
    (20, "last_trial", 500, 10, 12, 0, None, "throughput", True),
    (20, "last_trial", 40, 6, 12, 0, None, "latency", True)])
def test_decision_trace_matches_oracle(N, fallback, max_inner, max_outer, min_fd, min_sd, early_tol, kernel, cold):
    """Benchmark-family scenes (hard constraints active: F2 > 0, penalty growing); from a non-zero initial guess, or cold.  The
    oracle evaluates the L-BFGS operator in the form the kernel uses for the horizon (the Gram form at both compiled horizons)."""
    CAP, B = 240, 48
    cfg = make_cfg(N, solver_linesearch_fallback=fallback, solver_max_inner_iterations=max_inner,
                   solver_max_outer_iterations=max_outer)
    od = cfg.solver_dict(); od["lbfgs_gram"] = 1 if N in (20, 40) else 0
    ocfg = oracle.OracleConfig.from_dict(od)
    assert ocfg.ls_fallback == (1 if fallback == "half_step" else 0)
    bs = BatchSolver(cfg, library=variant_path("trace"), latency_batch=0 if kernel == "throughput" else None)
    bs.set_trace(CAP)
    sc = scenes.make_batch(cfg, B, n_dyn=8, seed=77 + N)
    u0 = None if cold else np.tile([0.6, 0.1], (B, N))
    res = bs.solve(sc["p"], u0)
    assert bool(bs.last_shape()["latency_kernel"]) == (kernel == "latency")
    tr = bs.read_trace(B)
    firsts, drifts, n_outer3, top_outer = [], [], 0, 0
    early = 0.0
    for b in range(B):
        _, ro, to, steps = oracle.solve_trace(ocfg, sc["p"][b], None if cold else u0[b], cap=CAP)
        tg = tr[b][~np.isnan(tr[b, :, 0])]
        assert len(tg) == min(steps, CAP) or res.num_inner_iterations[b] != ro["inner_iters"]
        n_outer3 += int(ro["outer_iters"] >= 3)
        fd = first_divergence(tg, to)
        top_outer = max(top_outer, int(to[:fd, 0].max()) if fd else 0)
        firsts.append(fd)
        # Scalars before the first discrete divergence.  Two float64 implementations of a descent iteration drift apart
        # geometrically (L-BFGS amplifies rounding differences), and identical decisions can be trivially identical (no
        # halving, no doubling, buffer full), so the scalar drift is measured on its own: `sd` = first step at which any
        # scalar differs by more than 1e-3 relative.  The first 10 steps must be tight.
        rel = np.zeros(fd)
        for f in SCALARS:
            a, o = tg[:fd, f], to[:fd, f]
            den = np.maximum(1e-300, np.maximum(np.abs(a), np.abs(o)))
            if f == 5:      # ||gamma fpr|| -> 0 at convergence: compare against the scale of the first steps
                den = np.maximum(den, 1e-6 * np.abs(to[0, f]))
            rel = np.maximum(rel, np.abs(a - o) / den)
        early = max(early, float(rel[:10].max()) if fd else 0.0)
        drifts.append(int(np.argmax(rel > SCALAR_TOL)) if (rel > SCALAR_TOL).any() else fd)
    firsts, drifts = np.array(firsts), np.array(drifts)
    hist = np.bincount(np.minimum(firsts // 25, 8), minlength=9)
    dhist = np.bincount(np.minimum(drifts // 25, 8), minlength=9)
    print(f"\n[N={N} {fallback} {max_inner}x{max_outer} {kernel}{' cold' if cold else ''}] first divergence of a discrete decision, per problem (bins of 25 steps, last = none in "
          f"{CAP}): {hist.tolist()}; median {np.median(firsts):.0f}, min {firsts.min()}; first step with a scalar off by > {SCALAR_TOL:g} "
          f"(or the discrete divergence, whichever comes first): {dhist.tolist()}; median {np.median(drifts):.0f}, min {drifts.min()}; "
          f"worst scalar rel. error in steps 0..9: {early:.2e}; problems with >= 3 outer iterations: {n_outer3}/{B}; highest outer "
          f"index matched {top_outer}")
    assert n_outer3 >= B // 2                       # the solves do go through several outer iterations
    assert np.median(firsts) >= min_fd              # typically dozens of identical decisions in a row
    if cold:
        assert firsts.min() >= 3                    # rounding-noise-limited Lipschitz estimate: see the parametrisation
    else:
        if max_inner * 3 <= CAP and N == 20:
            assert top_outer >= 2                   # ... and decisions were matched beyond the second penalty update
        assert early <= early_tol                   # the first 10 steps are tight
        assert firsts.min() >= 10                   # nobody diverges in the first steps
        assert np.median(drifts) >= min_sd and drifts.min() >= 10   # ... with scalars within 1e-3 over the first dozens of steps
    bs.close()


def test_linesearch_fallback_switch_changes_the_iteration_identically_on_both_sides():
    """Scenes where 10 halvings without acceptance occur: the two readings give different solutions, and each GPU
    reading tracks its oracle reading over the first iterations."""
    N, B = 20, 256
    sc = None
    out = {}
    for fb in ("last_trial", "half_step"):
        cfg = make_cfg(N, solver_linesearch_fallback=fb, solver_max_inner_iterations=60, solver_max_outer_iterations=3)
        if sc is None:
            sc = scenes.make_batch(cfg, B, n_dyn=8, seed=909)
        bs = BatchSolver(cfg)
        res = bs.solve(sc["p"], np.tile([0.6, 0.1], (B, N)))
        uo, _, ro, _ = oracle.solve_batch(oracle_cfg(cfg), sc["p"], np.tile([0.6, 0.1], (B, N)))
        out[fb] = (res, uo, ro)
        du = np.max(np.abs(res.solution - uo), axis=1)
        print(f"\n[{fb}] |du|inf GPU vs oracle after 3 x 60 iterations: median {np.median(du):.2e}, p90 {np.quantile(du, 0.9):.2e}")
        assert np.median(du) < 1e-3, (fb, np.median(du))
        assert np.mean(res.num_inner_iterations == ro["inner_iters"]) > 0.75
        bs.close()
    differs = np.max(np.abs(out["last_trial"][0].solution - out["half_step"][0].solution), axis=1) > 1e-6
    differs_o = np.max(np.abs(out["last_trial"][1] - out["half_step"][1]), axis=1) > 1e-6
    print(f"\nfallback reached in {differs.sum()}/{B} problems on the GPU, {differs_o.sum()}/{B} in the oracle")
    assert differs.sum() >= 1 and differs_o.sum() >= 1          # the switch is exercised
    assert np.mean(differs == differs_o) > 0.9                  # ... in the same problems


def test_penalty_stall_switch_changes_the_iteration_identically_on_both_sides():
    """The two readings of the ALM penalty-stall rule (yaml key solver_penalty_stall, MPCGPU_OPT_PENALTY_STALL; DESIGN.md section
    3).  "Passing" family: the acceleration constraints stay inactive in most problems (y+ = y = 0), so under "either" the penalty
    keeps its initial value 10 where "both" multiplies it by 5 per outer iteration whenever ||F2|| did not shrink.  Each GPU reading
    must follow ITS oracle reading (converged pairs within the north-star tolerance, the same outer-iteration counts and -- through
    them -- the same penalty path), and the switch must really change the iteration, in the same problems on both sides."""
    N, B = 20, 384
    out = {}
    sc = None
    for stall in ("either", "both"):
        cfg = make_cfg(N, solver_penalty_stall=stall)
        if sc is None:
            sc = scenes.make_batch(cfg, B, n_dyn=8, seed=4321, dyn_clearance=0.1, box_clearance=0.3)
        ocfg = oracle_cfg(cfg)
        assert ocfg.stall_rule == (1 if stall == "both" else 0)
        bs = BatchSolver(cfg)
        res = bs.solve(sc["p"])
        uo, _, ro, _ = oracle.solve_batch(ocfg, sc["p"])
        both = (res.status == 0) & (ro["status"] == 0)
        du = np.max(np.abs(res.solution - uo), axis=1)
        print(f"\n[{stall}] converged GPU {np.sum(res.status == 0)} / oracle {np.sum(ro['status'] == 0)} / both {both.sum()} of {B}; |du|inf on both: max "
              f"{du[both].max():.2e}; mean outer GPU {res.num_outer_iterations.mean():.2f} oracle {ro['outer_iters'].mean():.2f}; mean inner GPU "
              f"{res.num_inner_iterations.mean():.0f} oracle {ro['inner_iters'].mean():.0f}; final penalty (oracle) median {np.median(ro['penalty']):.0f}")
        assert both.sum() >= B // 5 and du[both].max() <= U_TOL
        assert np.array_equal(res.num_outer_iterations[both], ro["outer_iters"][both])
        assert np.mean((res.status == 0) == (ro["status"] == 0)) >= 0.9
        out[stall] = (res, uo, ro)
        bs.close()
    differs = np.max(np.abs(out["either"][0].solution - out["both"][0].solution), axis=1) > 1e-6
    differs_o = np.max(np.abs(out["either"][1] - out["both"][1]), axis=1) > 1e-6
    print(f"\nthe rule changes the answer in {differs.sum()}/{B} problems on the GPU, {differs_o.sum()}/{B} in the oracle")
    assert differs.sum() >= B // 10 and differs_o.sum() >= B // 10     # the switch is exercised ...
    assert np.mean(differs == differs_o) > 0.9                         # ... in the same problems
    # where neither reading ever raises the penalty the two are the same iteration: identical bits
    same = ~differs
    assert same.sum() >= B // 10
    assert np.array_equal(out["either"][0].num_inner_iterations[same], out["both"][0].num_inner_iterations[same])
    # under "either" the penalty of the oracle stays at its initial value in most problems of this family
    assert np.median(out["either"][2]["penalty"]) == 10.0 and np.median(out["both"][2]["penalty"]) > 10.0


@pytest.mark.parametrize("N,B", [(20, 512), (40, 128)])
def test_lbfgs_in_lds_build_is_bitwise_equal_to_the_product_build(N, B):
    """north_star words the layout as "L-BFGS memory staged in LDS"; the product build keeps it in the workspace record
    (measured).  The LDS variant must give the very same bits."""
    cfg = make_cfg(N)
    sc = scenes.make_batch(cfg, B, n_dyn=8, seed=31 + N, dyn_clearance=0.1, box_clearance=0.3)
    a = BatchSolver(cfg, latency_batch=0)           # throughput kernel on both sides (the latency kernel keeps S, Y in LDS anyway)
    b = BatchSolver(cfg, library=variant_path("lbfgs_lds"), latency_batch=0)
    ra, rb = a.solve(sc["p"]), b.solve(sc["p"])
    assert b.last_shape()["lds_bytes"] > a.last_shape()["lds_bytes"] + 2 * 10 * 2 * N * 8   # S and Y really are in LDS
    assert np.array_equal(ra.solution, rb.solution)
    assert np.array_equal(ra.num_inner_iterations, rb.num_inner_iterations)
    assert np.array_equal(ra.status, rb.status) and np.array_equal(ra.cost, rb.cost)
    a.close(); b.close()


@pytest.mark.parametrize("N,B,fam", [(20, 512, "passing"), (20, 256, "bench"), (40, 128, "passing")])
def test_one_call_site_build_is_bitwise_equal_to_the_product_build(N, B, fam):
    """The product kernel runs the PANOC steps in a loop of their own with separate call sites of the evaluation for the
    Lipschitz test and the line search (MPC_STEP_LOOP, mpc_kernels.hpp); rounds 1-3 ran ONE loop around ONE call site.  Same
    device functions on the same inputs: every output must agree bit for bit -- converged, cap-limited and half-way problems,
    evaluation counters included."""
    cfg = make_cfg(N)
    kw = dict(dyn_clearance=0.1, box_clearance=0.3) if fam == "passing" else {}
    sc = scenes.make_batch(cfg, B, n_dyn=8, seed=77 + N, **kw)
    a = BatchSolver(cfg, latency_batch=0)
    b = BatchSolver(cfg, library=variant_path("onesite"), latency_batch=0)
    ra, rb = a.solve(sc["p"]), b.solve(sc["p"])
    ea, eb = a.last_eval_counts(B), b.last_eval_counts(B)
    assert np.array_equal(ra.solution, rb.solution) and np.array_equal(ra.cost, rb.cost)
    assert np.array_equal(ra.num_inner_iterations, rb.num_inner_iterations)
    assert np.array_equal(ra.num_outer_iterations, rb.num_outer_iterations)
    assert np.array_equal(ra.status, rb.status)
    assert np.array_equal(np.asarray(ea), np.asarray(eb))
    assert len(set(ra.status.tolist())) >= (2 if fam == "passing" else 1)
    a.close(); b.close()


# (N_hor, family, config overrides, batch, floor of converged-on-both, |du| bound, floor of equal statuses, floor of equal inner counts)
# Measured with the oracle's two modes on the same scenes (CPU): N_hor = 20 "on_track": 505 of 512 converge in both forms, |du|inf
# 2.7e-7, 87 % with the same iteration count; N_hor = 40 (inner cap 5000, see the test above): 463 of 512, |du|inf 1.8e-6, and the
# iteration COUNTS differ by a median of 58 % -- with 80 unknowns the slow tail of the inner iteration is chaotic under rounding,
# the limit point is not -- so no count assertion there.
@pytest.mark.parametrize("N,family,cfgkw,B,min_both,du_tol,min_status,min_inner", [
    (20, "on_track", {}, 2048, 1900, 1e-6, 0.95, 0.8),
    (20, "passing", {}, 2048, 600, 2e-6, 0.95, 0.6),
    (40, "on_track", dict(solver_max_inner_iterations=5000), 2048, 1600, 1e-5, 0.9, None)])
def test_gram_form_of_the_lbfgs_operator_against_the_two_loop_build_on_the_gpu(N, family, cfgkw, B, min_both, du_tol, min_status, min_inner):
    """The product evaluates d = H (gamma fpr) in Gram form (PanocLbfgsGram); `make variants` keeps the two-loop recursion of the
    `lbfgs` crate (oracle/mpc_oracle.c:406-449) as libmpcgpu_twoloop.so.  Same operator, another summation order: the two builds
    solve the same batch on the GPU and must agree on the statuses, on the converged control sequences and -- rounding differences
    take a while to flip a decision -- on most iteration counts."""
    cfg = make_cfg(N, **cfgkw)
    sc = scenes.make_family(cfg, B, family, seed=99 + N)
    a = BatchSolver(cfg, latency_batch=0)
    b = BatchSolver(cfg, library=variant_path("twoloop"), latency_batch=0)
    ra, rb = a.solve(sc["p"]), b.solve(sc["p"])
    both = (ra.status == 0) & (rb.status == 0)
    du = np.max(np.abs(ra.solution - rb.solution), axis=1)
    same_status = np.mean(ra.status == rb.status)
    same_inner = np.mean(ra.num_inner_iterations[both] == rb.num_inner_iterations[both])
    print(f"\n[gram vs two-loop, GPU, N={N} {family} {cfgkw} B={B}] converged {np.sum(ra.status == 0)} / {np.sum(rb.status == 0)}, on both "
          f"{both.sum()}: |du|inf max {du[both].max():.2e}; same status {same_status:.4f}; same inner-iteration count among those {same_inner:.4f}")
    assert both.sum() >= min_both
    assert du[both].max() <= du_tol
    assert same_status >= min_status
    if min_inner is not None:
        assert same_inner >= min_inner
    a.close(); b.close()


def _two_loop_host(S, Y, r):
    """The `lbfgs` crate's recursion (oracle/mpc_oracle.c:406-449), newest pair first; S, Y lists oldest -> newest."""
    q = r.copy()
    al = []
    for s, y in zip(reversed(S), reversed(Y)):
        a = (s @ q) / (s @ y)
        al.append(a)
        q = q - a * y
    q = q * ((S[-1] @ Y[-1]) / (Y[-1] @ Y[-1]))
    for (s, y), a in zip(zip(S, Y), reversed(al)):
        be = (y @ q) / (s @ y)
        q = q + (a - be) * s
    return q


@pytest.mark.parametrize("N,m", [(20, 6), (20, 14), (40, 14)])
def test_gram_direction_against_the_two_loop_direction_on_recorded_pairs(N, m):
    """The L-BFGS operator alone (mpcgpu_debug_lbfgs_direction): the same recorded sequence of (u_j, gamma fpr_j) goes through the
    Gram form the product kernels use and through the two-loop recursion ON THE GPU, and through a host restatement of the
    recursion: d = H r agrees to 1e-11 relative -- with fewer pairs than the memory and with a wrapped ring."""
    cfg = make_cfg(N)
    B, n = 64, 2 * N
    rng = np.random.default_rng(N + m)
    U = np.zeros((B, m + 1, n)); R = np.zeros((B, m + 1, n))
    U[:, 0] = rng.normal(0, 0.3, (B, n)); R[:, 0] = rng.normal(0, 1e-2, (B, n))
    for b in range(B):
        Q = rng.normal(size=(n, n))
        M = np.diag(rng.uniform(0.5, 2.0, n)) + 0.05 * (Q @ Q.T) / n          # SPD: every pair has s'y > 0
        for j in range(1, m + 1):
            s = rng.normal(0, 1e-2, n)
            U[b, j] = U[b, j - 1] + s
            R[b, j] = R[b, j - 1] + 1e-2 * (M @ s)
    bs = BatchSolver(cfg)
    dg, dt, pairs = bs.debug_lbfgs_direction(U, R)
    bs.close()
    assert (pairs[:, 0] == min(m, 10)).all() and (pairs[:, 1] == min(m, 10)).all()      # every pair accepted by both forms
    worst_gt = worst_h = 0.0
    for b in range(B):
        S = [U[b, j] - U[b, j - 1] for j in range(1, m + 1)][-10:]
        Y = [R[b, j] - R[b, j - 1] for j in range(1, m + 1)][-10:]
        dh = _two_loop_host(S, Y, R[b, m])
        sc = np.max(np.abs(dh))
        worst_gt = max(worst_gt, np.max(np.abs(dg[b] - dt[b])) / sc)
        worst_h = max(worst_h, np.max(np.abs(dt[b] - dh)) / sc, np.max(np.abs(dg[b] - dh)) / sc)
    print(f"\n[L-BFGS direction, N={N}, {m} updates] Gram vs two-loop on the GPU: {worst_gt:.2e} relative; both vs the host recursion: {worst_h:.2e}")
    assert worst_gt <= 1e-11 and worst_h <= 1e-11


def test_solve_device_is_ordered_with_torch_work_on_the_same_stream_without_host_sync():
    """`p` is produced by a torch kernel right before solve_device and `u` is consumed by one right after, on torch's
    current stream (raw handle 0 for the default stream), with no host synchronisation in between."""
    import torch
    cfg = make_cfg(20)
    B = 256
    sc = scenes.make_batch(cfg, B, n_dyn=4, seed=5, dyn_clearance=0.1, box_clearance=0.3, v_init_range=(1.0, 1.2))
    bs = BatchSolver(cfg)
    ref = bs.solve(sc["p"])
    dev = torch.device("cuda:0")
    half = torch.from_numpy(0.5 * sc["p"]).to(dev)
    out = dict(u=torch.full((B, 40), 7.0, dtype=torch.float64, device=dev),
               cost=torch.empty(B, dtype=torch.float64, device=dev),
               status=torch.empty(B, dtype=torch.int32, device=dev))
    bs.reserve_shape(5, 0, 4, var_shape=False)                  # no count read-back: the call must not block
    for stream in (torch.cuda.current_stream(), torch.cuda.Stream()):
        with torch.cuda.stream(stream):
            big = torch.randn(4096, 4096, device=dev)
            for _ in range(10):                                  # keep the stream busy ahead of the producer of p
                big = big @ big * 1e-4
            p = half + half                                      # torch kernel producing p (exact: 0.5 p + 0.5 p)
            bs.solve_device(p, out, stream=stream.cuda_stream)
            total = out["u"].sum(dim=1)                          # torch kernel consuming u
        stream.synchronize()
        assert np.array_equal(out["u"].cpu().numpy(), ref.solution)
        assert torch.equal(total, torch.from_numpy(ref.solution).to(dev).sum(dim=1))
        out["u"].fill_(7.0)
    bs.close()


def test_reserved_shape_gives_identical_results_and_reports_problems_that_exceed_it():
    import torch
    cfg = make_cfg(20)
    B = 96
    sc = scenes.make_batch(cfg, B, n_dyn=6, n_other=2, seed=17)
    bs = BatchSolver(cfg)
    ref = bs.solve(sc["p"])
    dev = torch.device("cuda:0")
    p = torch.from_numpy(sc["p"]).to(dev)

    def run():
        out = dict(u=torch.empty(B, 40, dtype=torch.float64, device=dev), cost=torch.empty(B, dtype=torch.float64, device=dev),
                   status=torch.empty(B, dtype=torch.int32, device=dev), inner_it=torch.empty(B, dtype=torch.int32, device=dev))
        bs.solve_device(p, out, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        return {k: v.cpu().numpy() for k, v in out.items()}
    bs.reserve_shape()                                           # configured maxima, general tables
    full = run()
    assert bs.last_shape()["max_dyn"] == cfg.Ndynobs and bs.last_shape()["lds_bytes"] > 20000
    assert np.array_equal(full["u"], ref.solution) and np.array_equal(full["inner_it"], ref.num_inner_iterations)
    bs.reserve_shape(5, 2, 6, var_shape=False)                   # exact
    exact = run()
    assert np.array_equal(exact["u"], ref.solution)
    bs.reserve_shape(5, 2, 4, var_shape=False)                   # too small for every problem: reported, not solved
    small = run()
    assert np.all(small["status"] == 4) and np.all(np.isnan(small["cost"])) and np.all(small["u"] == 0.0)
    bs.release_shape()
    again = run()
    assert np.array_equal(again["u"], ref.solution)
    bs.close()
