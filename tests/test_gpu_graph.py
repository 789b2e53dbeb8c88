"""GPU (-m gpu): a reserved-shape solve captured into a hipGraph (torch.cuda.graph), replayed, and compared BITWISE with the
eager call -- the promise of include/mpcgpu.h (mpcgpu_reserve_shape + mpcgpu_reserve_batch: nothing is read back, allocated
or opted into inside the capture).  Also: a capture that would have to grow a buffer fails with a message instead of breaking
the capture, the axis-aligned reservation selects the same bits as the automatic rule, and a problem that breaks the
axis-aligned promise is reported (status 4) with every optional output overwritten."""
import numpy as np
import pytest
import torch

from conftest import make_cfg
from trajtrack_mpcndqn_rlboost_amd import BatchSolver, MpcGpuError, scenes

pytestmark = pytest.mark.gpu


def _out(B, n, dev):
    return dict(u=torch.empty(B, n, dtype=torch.float64, device=dev), cost=torch.empty(B, dtype=torch.float64, device=dev),
                status=torch.empty(B, dtype=torch.int32, device=dev), inner_it=torch.empty(B, dtype=torch.int32, device=dev),
                outer_it=torch.empty(B, dtype=torch.int32, device=dev), fpr=torch.empty(B, dtype=torch.float64, device=dev),
                f2norm=torch.empty(B, dtype=torch.float64, device=dev), y=torch.empty(B, n, dtype=torch.float64, device=dev),
                ms=torch.empty(B, dtype=torch.float64, device=dev))


@pytest.mark.parametrize("B", [1536, 4096, 6144])
def test_captured_solve_replays_bitwise(B):
    cfg = make_cfg(20, solver_max_inner_iterations=60, solver_max_outer_iterations=3)
    dev = torch.device("cuda", 0)
    sc = scenes.make_batch(cfg, B, n_dyn=8, seed=5, dyn_clearance=0.1, box_clearance=0.3)
    p = torch.from_numpy(sc["p"]).to(dev)
    # eager reference: the automatic rule (count read-back, axis-aligned kernel for these scenes)
    ref_solver = BatchSolver(cfg, latency_batch=0)
    ref = _out(B, 40, dev)
    ref_solver.solve_device(p, ref, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert ref_solver.last_shape()["max_dyn"] == 8

    bs = BatchSolver(cfg, latency_batch=0)
    bs.reserve_shape(max_static=5, max_fleet=0, max_dyn=8, var_shape=False, axis_aligned=True)
    out = _out(B, 40, dev)
    # a capture without pre-sized buffers must fail cleanly (-6), not break the capture
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with pytest.raises(MpcGpuError, match="reserve_batch"):
        with torch.cuda.graph(g, stream=side):
            bs.solve_device(p, out, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    bs.reserve_batch(B)
    if B > 4096:
        # more problems than are resident at once: after one eager call the handle holds evaluation counts, and the captured
        # launch carries the three ordering kernels (MPCGPU_OPT_ORDER) -- every replay re-sorts by the counts of the one before.
        # The hints are used only when they were written on the SAME stream (no ordering exists across streams): the eager call
        # goes to the stream that is captured afterwards.
        with torch.cuda.stream(side):
            bs.solve_device(p, out, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        bs.solve_device(p, out, stream=torch.cuda.current_stream().cuda_stream)
    assert bs.last_shape()["ordered"] == (B > 4096)
    for t in out.values():
        t.zero_()
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    for k in ("u", "cost", "status", "inner_it", "outer_it", "fpr", "f2norm", "y"):
        assert torch.equal(out[k], ref[k]), k
    assert int((out["status"] == 0).sum()) > 0
    bs.close(); ref_solver.close()


def test_broken_axis_promise_is_reported_and_overwrites_every_output():
    cfg = make_cfg(20, solver_max_inner_iterations=40, solver_max_outer_iterations=2)
    dev = torch.device("cuda", 0)
    B = 2048
    sc = scenes.make_batch(cfg, B, n_dyn=4, seed=9)
    p = sc["p"].copy()
    # problem 7: rotate its first dynamic row (angle != 0 on every step)
    N = cfg.N_hor
    od0 = 18 + 4 * N + 3 * N * cfg.Nother + cfg.Nstcobs * 12
    p[7, od0 + 4:od0 + 6 * N:6] = 0.3
    pt = torch.from_numpy(p).to(dev)
    bs = BatchSolver(cfg, latency_batch=0)
    bs.reserve_shape(var_shape=False, axis_aligned=True)
    out = _out(B, 40, dev)
    for k in ("fpr", "f2norm", "y", "ms"):
        out[k].fill_(123.0)
    bs.solve_device(pt, out, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    st = out["status"].cpu().numpy()
    assert st[7] == 4 and (np.delete(st, 7) != 4).all()
    assert np.isnan(out["cost"][7].item()) and np.isnan(out["fpr"][7].item()) and np.isnan(out["f2norm"][7].item())
    assert torch.isnan(out["y"][7]).all() and out["ms"][7].item() == 0.0 and (out["u"][7] == 0).all()
    # the same batch under the general reservation: problem 7 is solved like everyone else
    bs.reserve_shape(var_shape=True)
    bs.solve_device(pt, out, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert (out["status"].cpu().numpy() != 4).all()
    bs.close()


@pytest.mark.parametrize("B", [256, 768])
def test_small_batches_capture_without_a_reservation(B):
    """The latency range (up to four problems per compute unit) needs no mpcgpu_reserve_shape to be captured: inside a capture
    the whole range takes the one-launch form (compaction fused, tables for the configured maxima, nothing read back) -- also
    above two problems per compute unit, where the eager rule would read the batch's row counts back for its mid-batch form."""
    cfg = make_cfg(20, solver_max_inner_iterations=60, solver_max_outer_iterations=3)
    dev = torch.device("cuda", 0)
    sc = scenes.make_batch(cfg, B, n_dyn=6, seed=15, dyn_clearance=0.1, box_clearance=0.3)
    p = torch.from_numpy(sc["p"]).to(dev)
    ref_solver = BatchSolver(cfg, latency_batch=1024)      # the whole latency range, mid-range form included (the default ends at 512)
    ref = _out(B, 40, dev)
    ref_solver.solve_device(p, ref, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert ref_solver.last_shape()["latency_kernel"]
    bs = BatchSolver(cfg, latency_batch=1024)
    bs.reserve_batch(B)
    out = _out(B, 40, dev)
    bs.solve_device(p, out, stream=torch.cuda.current_stream().cuda_stream)     # one eager call: LDS opt-in of the kernel
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        bs.solve_device(p, out, stream=torch.cuda.current_stream().cuda_stream)
    assert bs.last_shape()["latency_kernel"]
    for t in out.values():
        t.zero_()
    g.replay()
    torch.cuda.synchronize()
    for k in ("u", "cost", "status", "inner_it", "outer_it", "fpr", "f2norm", "y"):
        assert torch.equal(out[k], ref[k]), k
    bs.close(); ref_solver.close()
