"""GPU (-m gpu): the tracker harness and the hybrid loop's reference generator ON THE DEVICE (csrc/mpc_tracker.hpp,
csrc/trackgpu.hip) against their host forms, which the reference's own traces pin
(tests/test_tracker_harness.py, tests/test_hybrid_logic.py):

* the assembly kernel writes the solver's compact record DIRECTLY from the tracker's arrays -- bitwise what the compaction
  kernel makes of BatchedTracker.assemble()'s padded vectors (the speed reference goes through hypot, whose last bit differs
  between libm and the device library: that one block is compared to 1 ulp);
* the window search picks the same indices and rows;
* a closed loop of DeviceTracker stays on BatchedTracker's trajectory;
* the DQN proposal rollout and the batched HintSwitcher agree with dqn.rl_reference / hybrid.BatchedHintSwitcher."""
import importlib

import numpy as np
import pytest
import torch

from conftest import make_cfg
from trajtrack_mpcndqn_rlboost_amd import BatchSolver
from trajtrack_mpcndqn_rlboost_amd.batched_tracker import BatchedTracker
from trajtrack_mpcndqn_rlboost_amd.device_tracker import DeviceTracker

pytestmark = pytest.mark.gpu


def _setup(cfg, B, rng, host, dev):
    for i in range(B):
        y = rng.uniform(2, 5)
        path = [(0.6, y), (rng.uniform(4, 8), rng.uniform(2, 6)), (rng.uniform(9, 15), 3.5)][:(2 if i % 3 == 0 else 3)]
        polys = [[(6.7, 2.2), (9.3, 2.2), (9.3, 4.8), (6.7, 4.8)]] * (i % 3)
        for t in (host, dev):
            t.initialization(i, np.array([0.6, y, 0.0]), np.array([path[-1][0], path[-1][1], 0.0]), path, "work")
            t.update_static_constraints(i, polys)


@pytest.mark.parametrize("N", [20, 40])
def test_device_assembly_writes_the_record_the_compaction_kernel_makes_of_the_host_assembly(N):
    cfg = make_cfg(N)
    B = 96
    solver = BatchSolver(cfg)
    host = BatchedTracker(cfg, B, solver=solver)
    dev = DeviceTracker(cfg, B, solver=solver)
    rng = np.random.default_rng(0)
    _setup(cfg, B, rng, host, dev)
    dev.view()                                                    # uploads the set-up
    off = cfg.offsets()
    for trial in range(12):
        # somewhere along (late in the run: at the end of) the reference, random last actions, obstacle and fleet rows
        idx0 = host.idx_ref.copy()
        for i in range(B):
            k = min(host._ref_len[i] - 1, idx0[i] + rng.integers(0, 4) + (trial > 8) * 400)
            host.states[i, :2] = host._ref[i, k, :2] + rng.normal(0, 0.05, 2)
            host.states[i, 2] = rng.uniform(-3, 3)
        host.last_actions[:] = rng.normal(size=host.last_actions.shape)
        n_dyn = rng.integers(0, cfg.Ndynobs + 1, B)
        dynp = rng.normal(size=(B, cfg.Ndynobs, N, 6))
        dynp[..., 2:4] = np.abs(dynp[..., 2:4]) + 0.2
        shape_const = rng.random(B) < 0.5                         # rows that keep their shape over the horizon / axis-aligned ones
        dynp[shape_const, :, :, 2:] = dynp[shape_const, :, :1, 2:]
        dynp[(rng.random(B) < 0.3)[:, None] & np.ones((1, cfg.Ndynobs), bool), :, 4] = 0.0
        dynp[np.arange(cfg.Ndynobs)[None, :] >= n_dyn[:, None]] = 0.0
        host.dyn_constraints[:] = dynp.reshape(B, -1)
        host.other_robot_states[:] = rng.normal(size=host.other_robot_states.shape) * (rng.random((B, 1)) < 0.5)
        dev.states.copy_(torch.from_numpy(host.states)); dev.last_actions.copy_(torch.from_numpy(host.last_actions))
        dev.idx_ref.copy_(torch.from_numpy(idx0.astype(np.int32)))
        dev.set_dynamic_constraints(dynp); dev.set_other_robot_states(host.other_robot_states)
        # window search
        refs_h = host.local_refs()
        refs_d = dev.local_refs()
        torch.cuda.synchronize()
        assert np.array_equal(dev.idx_ref.cpu().numpy(), host.idx_ref)
        assert np.array_equal(refs_d.cpu().numpy(), refs_h)
        # records
        P = host.assemble("work", refs_h)
        solver.debug_prep(P)
        want, rec = solver.debug_workspace(B)
        solver.debug_tracker_assemble(dev.view(), refs_d)
        got, _ = solver.debug_workspace(B)
        vref = slice(64, 64 + N)                                   # HDR doubles of header, then the speed references
        assert np.array_equal(np.delete(got[:, :rec], np.r_[vref], axis=1), np.delete(want[:, :rec], np.r_[vref], axis=1),
                              equal_nan=True)
        assert np.allclose(got[:, vref], want[:, vref], rtol=4e-16, atol=0)
        assert np.array_equal(got[:, vref] == cfg.lin_vel_max * cfg.high_speed, want[:, vref] == cfg.lin_vel_max * cfg.high_speed)
    assert (host.idx_ref + N >= host._ref_len).any()               # the tail padding and the near-goal speed rule were exercised
    assert (P[:, off["vref"]] < cfg.lin_vel_max * cfg.high_speed).any()
    solver.close()


def test_device_tracker_closed_loop_follows_the_host_tracker():
    cfg = make_cfg(20)
    B = 48
    rng = np.random.default_rng(3)
    host = BatchedTracker(cfg, B)
    dev = DeviceTracker(cfg, B)
    for i in range(B):                                              # free corridor: the solves converge
        y = rng.uniform(3, 6)
        path = [(0.6, y), (14.0, y + rng.uniform(-1, 1))]
        for t in (host, dev):
            t.initialization(i, np.array([0.6, y, rng.uniform(-0.2, 0.2)]) if t is host else host.states[i], np.array([path[-1][0], path[-1][1], 0.0]), path)
    worst = 0.0
    for tick in range(12):
        a_h, pred_h, cost_h = host.step()
        out = dev.step()
        torch.cuda.synchronize()
        # same states in -> bitwise the same records -> bitwise the same solves (converged or not)
        assert np.array_equal(out["u"].cpu().numpy(), host.last_result.solution), tick
        assert np.array_equal(out["status"].cpu().numpy(), host.last_result.status)
        assert np.array_equal(out["cost"].cpu().numpy(), cost_h)
        assert np.array_equal(dev.idx_ref.cpu().numpy(), host.idx_ref)
        # the rollouts use the device's sin / cos: last-bit differences
        d = np.abs(dev.states.cpu().numpy() - host.states).max()
        worst = max(worst, d)
        assert d < 1e-12
        assert np.abs(out["actions"].cpu().numpy() - a_h).max() == 0.0
        assert np.abs(dev.pred_states.cpu().numpy() - pred_h).max() < 1e-11
        assert np.array_equal(dev.active.cpu().numpy().astype(bool), host.active)
        assert np.array_equal(dev.last_actions.cpu().numpy(), host.last_actions)
        # keep the two loops on the same state so that every tick compares like with like
        dev.states.copy_(torch.from_numpy(host.states))
    assert (host.last_result.status == 0).mean() > 0.5                 # the loop reached ticks that converge
    print(f"\n[device tracker] 12 ticks x {B} robots: bitwise equal solves; max |state_device - state_host| {worst:.2e}")


def test_rl_reference_and_hint_switch_kernels_match_the_host_forms():
    dqn = importlib.import_module("trajtrack_mpcndqn_rlboost_amd.dqn")
    hybrid = importlib.import_module("trajtrack_mpcndqn_rlboost_amd.hybrid")
    cfg = make_cfg(20)
    bs = BatchSolver(cfg)
    rng = np.random.default_rng(5)
    B, N, O, V = 512, 20, 6, 8
    dev = torch.device("cuda", 0)
    # ---- proposal rollout
    agent = np.concatenate([rng.uniform(0, 10, (B, 2)), rng.uniform(-3, 3, (B, 1)), rng.uniform(-0.5, 1.5, (B, 1)),
                            rng.uniform(-1, 1, (B, 1)), rng.normal(size=(B, 3))], axis=1)
    action = rng.integers(0, 9, B)
    want, _ = dqn.rl_reference(agent[:, :5], action, cfg.ts, steps=20, ref_speed=1.0)
    rl_ref = torch.empty(B, 20, 2, dtype=torch.float64, device=dev)
    lim = (dqn.ACCELERATION_MAX, dqn.ACCELERATION_MIN, dqn.ANGULAR_ACCELERATION_MAX, dqn.ANGULAR_ACCELERATION_MIN,
           dqn.SPEED_MIN, dqn.SPEED_MAX, dqn.ANGULAR_VELOCITY_MIN, dqn.ANGULAR_VELOCITY_MAX)
    st = torch.cuda.current_stream().cuda_stream      # ordered with the torch copies around the calls
    bs.rl_reference(torch.from_numpy(agent).to(dev), torch.from_numpy(action).to(dev), cfg.ts, 20, 1.0, lim, rl_ref, stream=st)
    assert np.abs(rl_ref.cpu().numpy() - want).max() < 1e-12
    # ---- switcher: 25 ticks of random geometry, state carried on both sides
    sw = hybrid.BatchedHintSwitcher(B, 10, 2, 10)
    on_d = torch.zeros(B, dtype=torch.uint8, device=dev)
    cnt_d = torch.zeros(B, dtype=torch.int32, device=dev)
    chosen = torch.empty(B, N, 3, dtype=torch.float64, device=dev)
    seen_on = seen_off = 0
    for tick in range(25):
        centres = rng.uniform(0, 12, (B, O, 1, 2))
        ang = np.sort(rng.uniform(0, 2 * np.pi, (B, O, V)), axis=2)
        rad = rng.uniform(0.3, 2.0, (B, O, V))
        poly = centres + np.stack([rad * np.cos(ang), rad * np.sin(ang)], axis=-1)
        poly[:, :, 6:] = poly[:, :, 5:6]                               # padded rings (repeated last vertex)
        valid = rng.random((B, O)) < 0.7
        pos = rng.uniform(0, 12, (B, 3))
        original = np.concatenate([rng.uniform(0, 12, (B, N, 2)), rng.uniform(-3, 3, (B, N, 1))], axis=2)
        original[:, :, :2] = np.where(rng.random((B, 1, 1)) < 0.5, centres[:, 0] + rng.normal(0, 0.2, (B, N, 2)), original[:, :, :2])
        live = rng.random(B) < 0.9
        prev = sw.switch_on.copy()
        on_h = sw.switch(pos[:, :2], original, poly, valid, live)
        bs.hint_switch(torch.from_numpy(poly).to(dev), torch.from_numpy(valid.astype(np.uint8)).to(dev), torch.from_numpy(pos).to(dev),
                       torch.from_numpy(original).to(dev), rl_ref, torch.from_numpy(live.astype(np.uint8)).to(dev), (10, 2, 10),
                       on_d, cnt_d, chosen, stream=st)
        assert np.array_equal(on_d.cpu().numpy().astype(bool), on_h)
        assert np.array_equal(cnt_d.cpu().numpy(), sw.detach_cnt)
        use = on_h & live
        want_c = np.where(use[:, None, None], np.concatenate([want[:, :N], original[..., 2:3]], axis=2), original)
        assert np.array_equal(chosen.cpu().numpy()[~use], want_c[~use])
        assert np.abs(chosen.cpu().numpy() - want_c).max() < 1e-12
        seen_on += int((on_h & ~prev).sum()); seen_off += int((~on_h & prev).sum())
    assert seen_on > 50 and seen_off > 5                                # both transitions were exercised
    bs.close()


def test_set_up_calls_in_the_middle_of_a_run_touch_only_their_robot():
    """`update_static_constraints` may be called at any time (src/interface_mpc.py:60-63) and re-planning one robot must not move
    the others: after some ticks, robot 0 gets a new map and robot 1 a new plan; every other robot keeps the state, reference
    index, last action and activity flag the device has advanced it to, robot 1 alone starts over."""
    cfg = make_cfg(20, solver_max_inner_iterations=30, solver_max_outer_iterations=2)
    B = 12
    dev = DeviceTracker(cfg, B)
    host = BatchedTracker(cfg, B, solver=dev.solver)
    _setup(cfg, B, np.random.default_rng(3), host, dev)
    for _ in range(5):
        dev.step()
    torch.cuda.synchronize()
    before = {k: getattr(dev, k).clone() for k in ("states", "idx_ref", "last_actions", "active", "stc", "goals", "ref_len")}
    assert float((before["states"][:, 0] - 0.6).abs().min()) > 1e-3          # everybody has moved
    dev.update_static_constraints(0, [[(1.0, 1.0), (2.0, 1.0), (2.0, 2.0), (1.0, 2.0)]])
    long_path = [(0.5, 3.0)] + [(0.5 + 3.0 * j, 3.0 + (j % 2)) for j in range(1, 12)]        # longer than any reference so far
    dev.initialization(1, np.array([0.5, 3.0, 0.1]), np.array([long_path[-1][0], long_path[-1][1], 0.0]), long_path, "work")
    dev.view()
    torch.cuda.synchronize()
    others = [i for i in range(B) if i != 1]
    for k in ("states", "idx_ref", "last_actions", "active", "goals", "ref_len"):
        assert torch.equal(getattr(dev, k)[others], before[k][others]), k
    assert torch.equal(dev.stc[1:], before["stc"][1:]) and not torch.equal(dev.stc[0], before["stc"][0])
    assert torch.equal(dev.states[1].cpu(), torch.tensor([0.5, 3.0, 0.1], dtype=torch.float64))
    assert int(dev.idx_ref[1]) == 0 and int(dev.active[1]) == 1 and float(dev.last_actions[1].abs().sum()) == 0.0
    assert int(dev.ref_len[1]) > int(before["ref_len"].max()) and dev.ref.shape[1] == int(dev.ref_len[1])
    out = dev.step()                                                           # and the run goes on
    torch.cuda.synchronize()
    assert int((out["status"] >= 0).sum()) == B
