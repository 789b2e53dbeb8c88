"""GPU (-m gpu): the balanced walk of the dynamic rows at N_hor = 40 (round 4, csrc/mpc_kernels.hpp eval_point).

Steps 0-23 of a 40-step horizon have two item lanes, steps 24-39 one; walking the 8 dynamic rows step by step the single-lane
steps take 8 trips while the others finish in 4.  The product lets every lane make 5 trips: the lanes of the two-lane steps take
rows 5-7 of the single-lane steps as foreign items and add their gradients to that step's accumulator in LDS.  That is another
summation order for those steps, so the product is not bitwise the row walk of rounds 1-3 (kept as libmpcgpu_rowwalk40.so):
  * one evaluation (psi, f, grad psi, F1, F2) agrees to rounding, with hard constraints active;
  * whole solves agree like two float64 implementations do: same statuses, converged control sequences within 1e-6;
  * every kernel that shares eval_point moved together: the latency kernel stays BITWISE the throughput kernel (tests/test_gpu_latency.py),
    and the row-walk build is bitwise its own latency kernel too."""
import numpy as np
import pytest

from conftest import make_cfg
from trajtrack_mpcndqn_rlboost_amd import BatchSolver, scenes
from trajtrack_mpcndqn_rlboost_amd.solver import variant_path

pytestmark = pytest.mark.gpu
ROW = variant_path("rowwalk40")


@pytest.mark.parametrize("n_dyn", [8, 5, 15])
def test_one_evaluation_agrees_with_the_row_walk_to_rounding(n_dyn):
    cfg = make_cfg(40)
    B = 512
    sc = scenes.make_family(cfg, B, "benchmark", n_dyn=n_dyn, seed=20 + n_dyn)
    rng = np.random.default_rng(1)
    u = np.stack([rng.uniform(-0.4, 1.4, (B, 40)), rng.uniform(-0.5, 0.5, (B, 40))], axis=2).reshape(B, 80)
    y = rng.normal(0, 1.0, (B, 80))
    a, b = BatchSolver(cfg), BatchSolver(cfg, library=ROW)
    ga = a.cost_grad(u, sc["p"], c=np.full(B, 250.0), y=y)
    gb = b.cost_grad(u, sc["p"], c=np.full(B, 250.0), y=y)
    assert float(np.abs(ga["F2"]).max()) > 0.0                        # hard constraints are violated: both item passes run
    assert np.array_equal(ga["F2"], gb["F2"]) and np.array_equal(ga["F1"], gb["F1"])     # hinge row sums: the same atomics
    for k in ("psi", "f"):
        assert np.max(np.abs(ga[k] - gb[k]) / np.maximum(1.0, np.abs(gb[k]))) <= 1e-13, k
    scale = np.maximum(1.0, np.max(np.abs(gb["grad"]), axis=1, keepdims=True))
    assert np.max(np.abs(ga["grad"] - gb["grad"]) / scale) <= 1e-13
    differs = np.mean(np.any(ga["grad"] != gb["grad"], axis=1))
    print(f"\n[balanced walk, {n_dyn} dynamic rows] gradients differ in the last bits for {differs:.2f} of the problems")
    a.close(); b.close()


def test_whole_solves_agree_with_the_row_walk():
    cfg = make_cfg(40, solver_max_inner_iterations=5000)
    B = 2048
    sc = scenes.make_family(cfg, B, "on_track", seed=139)
    a, b = BatchSolver(cfg, latency_batch=0), BatchSolver(cfg, latency_batch=0, library=ROW)
    ra, rb = a.solve(sc["p"]), b.solve(sc["p"])
    both = (ra.status == 0) & (rb.status == 0)
    du = np.max(np.abs(ra.solution - rb.solution), axis=1)
    print(f"\n[balanced vs row walk, N=40 on_track] converged {np.sum(ra.status == 0)} / {np.sum(rb.status == 0)}, on both {both.sum()}: "
          f"|du|inf max {du[both].max():.2e}; same status {np.mean(ra.status == rb.status):.4f}")
    assert both.sum() >= 1600 and du[both].max() <= 1e-5 and np.mean(ra.status == rb.status) >= 0.9
    a.close(); b.close()


def test_the_row_walk_build_is_bitwise_its_own_latency_kernel():
    cfg = make_cfg(40, solver_max_inner_iterations=80, solver_max_outer_iterations=3)
    B = 96
    sc = scenes.make_family(cfg, B, "benchmark", seed=7)
    t = BatchSolver(cfg, library=ROW, latency_batch=0)
    l = BatchSolver(cfg, library=ROW)
    rt, rl = t.solve(sc["p"]), l.solve(sc["p"])
    assert l.last_shape()["latency_kernel"] and not t.last_shape()["latency_kernel"]
    for k in ("solution", "cost", "status", "num_inner_iterations", "num_outer_iterations"):
        assert np.array_equal(getattr(rt, k), getattr(rl, k)), k
    t.close(); l.close()
