"""GPU (-m gpu): linear centre tables (round-4 EXPERIMENT, csrc/mpc_kernels.hpp prep_problem / dyn_item<.., LIN>; compiled only
into the variant build libmpcgpu_linear40.so: it is bitwise equal and measured slower, profiles/r04_linear_tables_ab.txt).

A constant-velocity prediction (est_dyn_obs_positions, /root/reference/src/main.py:77-85) puts the centres of a dynamic row on a
straight line up to rounding: centre k = fma(n_k, u, fma(d, k, c0)) with a 16-bit integer n_k gives the stored double back bit for
bit.  At N_hor = 40 the table shrinks from 5 to 1.7 KB per problem and the 128-register build of the solve kernel keeps 16 instead
of 12 problems resident per compute unit.  The compression is lossless, so every output must be BITWISE that of the stored-centre
kernel; a problem with a row that does not fit (curved predictions) is picked up by a second launch from the stored centres."""
import numpy as np
import pytest

from conftest import make_cfg
from trajtrack_mpcndqn_rlboost_amd import BatchSolver, scenes
from trajtrack_mpcndqn_rlboost_amd.feeders import constant_velocity_prediction
from trajtrack_mpcndqn_rlboost_amd.solver import variant_path

LIN = variant_path("linear40")      # built on the row walk of the dynamic rows (as measured): compared with that build
ROW = variant_path("rowwalk40")

pytestmark = pytest.mark.gpu


def _same(a, b):
    for k in ("solution", "cost", "status", "num_inner_iterations", "num_outer_iterations", "last_problem_norm_fpr", "f2_norm",
              "lagrange_multipliers"):
        assert np.array_equal(getattr(a, k), getattr(b, k), equal_nan=True), k


@pytest.mark.parametrize("family", ["passing", "benchmark"])
def test_linear_tables_give_the_bits_of_the_stored_centres(family):
    cfg = make_cfg(40, solver_max_inner_iterations=300, solver_max_outer_iterations=4)
    B = 6144                                          # more than the 3072 problems the three-wavefront build keeps resident
    sc = scenes.make_family(cfg, B, family, seed=404)
    lin = BatchSolver(cfg, latency_batch=0, order="as_given", library=LIN)
    ref = BatchSolver(cfg, latency_batch=0, order="as_given", library=ROW)
    ra, rb = lin.solve(sc["p"]), ref.solve(sc["p"])
    sa, sb = lin.last_shape(), ref.last_shape()
    assert sa["linear"] and sa["waves_per_simd"] == 4 and sa["lds_bytes"] <= 10240
    assert not sb["linear"] and sb["waves_per_simd"] == 3 and sb["lds_bytes"] > 12000
    _same(ra, rb)
    ea, eb = lin.last_eval_counts(B), ref.last_eval_counts(B)
    assert np.array_equal(ea[0], eb[0]) and np.array_equal(ea[1], eb[1])
    # a batch that fits the three-wavefront build anyway keeps the stored centres (nothing to gain)
    small = lin.solve(sc["p"][:2048])
    assert not lin.last_shape()["linear"] and lin.last_shape()["waves_per_simd"] == 3
    assert np.array_equal(small.solution, ra.solution[:2048])
    lin.close(); ref.close()


def test_the_reference_feeder_is_linear_and_a_curved_row_is_not():
    cfg = make_cfg(40, solver_max_inner_iterations=60, solver_max_outer_iterations=3)
    N, B = 40, 4096
    off = cfg.offsets()
    sc = scenes.make_family(cfg, B, "passing", seed=11)
    p = sc["p"].copy()
    # rows 0..3 of every problem: what est_dyn_obs_positions produces (current + (k + 1) * (current - last))
    rng = np.random.default_rng(3)
    cur = p[:, off["od"]:off["od"] + 2][:, None, :] + rng.normal(0, 0.5, (B, 4, 2))
    last = cur - rng.uniform(-0.3, 0.3, (B, 4, 2))
    p[:, off["od"]:off["od"] + 4 * 6 * N] = constant_velocity_prediction(last, cur, steps=N).reshape(B, -1)
    bs = BatchSolver(cfg, latency_batch=0, library=LIN)
    ref = BatchSolver(cfg, latency_batch=0, library=ROW)
    ra, rb = bs.solve(p), ref.solve(p)
    assert bs.last_shape()["linear"]
    _same(ra, rb)
    # some problems get a row on a parabola: the linear-table launch leaves them to the pick-up launch (stored centres), the others
    # do not notice; the same through the device-pointer entry under a reservation (nothing read back)
    q = p.copy()
    k = np.arange(N)
    curved = [17, 18, 2000, 4095]
    for j in curved:
        q[j, off["od"] + 6 * k] += 1e-3 * (k - 20.0) ** 2
    rc = bs.solve(q)
    assert bs.last_shape()["linear"] and bs.last_shape()["waves_per_simd"] == 4
    rd = ref.solve(q)
    _same(rc, rd)
    assert np.array_equal(np.delete(rc.solution, curved, 0), np.delete(ra.solution, curved, 0))
    import torch
    dev = torch.device("cuda", 0)
    out = dict(u=torch.empty(B, 2 * N, dtype=torch.float64, device=dev), cost=torch.empty(B, dtype=torch.float64, device=dev),
               status=torch.empty(B, dtype=torch.int32, device=dev))
    bs.reserve_shape(max_static=5, max_fleet=0, max_dyn=8, var_shape=False, axis_aligned=True)
    bs.solve_device(torch.from_numpy(q).to(dev), out, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert bs.last_shape()["linear"]
    assert np.array_equal(out["u"].cpu().numpy(), rd.solution) and np.array_equal(out["status"].cpu().numpy(), rd.status)
    bs.close(); ref.close()


def test_cost_and_gradient_through_the_linear_tables_are_bitwise_the_stored_ones():
    """The test hook has no pick-up launch: it goes through the linear tables only when EVERY problem of the batch fits them (a
    coordinate within ~1/4000 of the largest one of zero does not), so small batches are tried until one does."""
    cfg = make_cfg(40)
    B = 48
    rng = np.random.default_rng(0)
    a = BatchSolver(cfg, library=LIN)
    b = BatchSolver(cfg, library=ROW)
    checked = 0
    for seed in range(2, 14):
        sc = scenes.make_family(cfg, B, "benchmark", seed=seed)
        u = rng.uniform(-0.4, 1.2, (B, 80))
        y = rng.normal(0, 1.0, (B, 80))
        ga = a.cost_grad(u, sc["p"], c=np.full(B, 50.0), y=y)
        if not a.last_shape()["linear"]:
            continue
        gb = b.cost_grad(u, sc["p"], c=np.full(B, 50.0), y=y)
        assert not b.last_shape()["linear"]
        for k in ("psi", "f", "grad", "F1", "F2"):
            assert np.array_equal(ga[k], gb[k]), k
        assert float(np.abs(ga["F2"]).max()) > 0.0        # hard constraints are active in this family: the hard branch is exercised
        checked += 1
    assert checked >= 3, checked
    a.close(); b.close()
