"""CPU: host-side logic -- config surface, parameter layout, scene generator, C-ABI export table, and the
"fail loudly without a GPU" behaviour of the product path."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, make_cfg
import importlib

from trajtrack_mpcndqn_rlboost_amd import MpcConfig, scenes

solver_mod = importlib.import_module("trajtrack_mpcndqn_rlboost_amd.solver")  # (the package attribute `solver` is the plugin factory)


def test_config_surface_matches_reference_keys(meta):
    cfg = MpcConfig()
    # keys the reference reads from the YAML (config/mpc_default.yaml:7-55)
    for key in ("vehicle_width", "vehicle_margin", "social_margin", "lin_vel_min", "lin_vel_max", "lin_acc_min",
                "lin_acc_max", "ang_vel_max", "ang_acc_max", "full_speed", "high_speed", "medium_speed",
                "low_speed", "ts", "N_hor", "action_steps", "lin_vel_penalty", "lin_acc_penalty",
                "ang_vel_penalty", "ang_acc_penalty", "qrpd", "qpos", "qvel", "qtheta", "qpN", "qthetaN", "nu",
                "ns", "nq", "Nother", "Nstcobs", "nstcobs", "Ndynobs", "ndynobs", "build_type",
                "build_directory", "bad_exit_codes", "optimizer_name"):
        assert hasattr(cfg, key), key
    assert cfg.num_params == 2658 and cfg.num_decision == 40
    assert MpcConfig(N_hor=40).num_params == 5178
    assert cfg.bad_exit_codes == ["NotConvergedIterations", "NotConvergedOutOfTime"]
    for name, opt in (("mpc_default.yaml", "navi_default"), ("mpc_longiter.yaml", "navi_longiter"),
                      ("mpc_test.yaml", "navi_test")):
        c = MpcConfig(os.path.join(ROOT, "config", name))
        assert c.optimizer_name == opt and c.N_hor == 20
    for k, v in meta["N20"]["yaml"].items():
        assert getattr(cfg, k) == v


def test_parameter_offsets():
    cfg = MpcConfig()
    off = cfg.offsets()
    assert off == dict(s=0, q=8, r=18, vref=78, c=98, os=698, od=818, qstc=2618, qdyn=2638, end=2658)


def test_scene_generator_is_seeded_and_well_formed():
    cfg = MpcConfig()
    a = scenes.make_batch(cfg, 32, n_dyn=8, seed=11)
    b = scenes.make_batch(cfg, 32, n_dyn=8, seed=11)
    c = scenes.make_batch(cfg, 32, n_dyn=8, seed=12)
    assert a["p"].shape == (32, cfg.num_params) and a["p"].dtype == np.float64
    assert np.array_equal(a["p"], b["p"]) and not np.array_equal(a["p"], c["p"])
    off = cfg.offsets()
    N = cfg.N_hor
    p = a["p"]
    assert np.all(np.isfinite(p))
    # weights of the 'work' mode (trajectory_generator.py:129-130)
    assert np.allclose(p[:, off["q"]:off["q"] + 10], [0, 10, 0, 0, 0, 0, 0, 100, 10, 20])
    # reference points are 0.24 m apart
    ref = p[:, off["r"]:off["r"] + 3 * N].reshape(32, N, 3)
    d = np.hypot(np.diff(ref[:, :, 0], axis=1), np.diff(ref[:, :, 1], axis=1))
    assert np.allclose(d, 0.24)
    # 8 active dynamic rows with radius 1.6, the remaining 7 rows are zero padding
    od = p[:, off["od"]:off["od"] + cfg.Ndynobs * 6 * N].reshape(32, cfg.Ndynobs, N, 6)
    assert np.all(od[:, :8, :, 2] == 1.6) and np.all(od[:, 8:] == 0.0)
    # the first predicted obstacle position is clear of the robot
    assert np.all(np.hypot(od[:, :8, 0, 0] - p[:, None, 0], od[:, :8, 0, 1] - p[:, None, 1]) >= 1.6 + 0.69)
    assert np.all(p[:, off["qdyn"]:] == 1e3)


def test_rect_halfspaces_match_reference_representation():
    """polygon_halfspace_representation of the reference (util/utils_geo.py:33-59) on its own rectangle probe
    (fixture generated from the reference; rows may be ordered differently)."""
    fx = np.load(os.path.join(ROOT, "tests", "golden", "halfspace.npz"))
    poly, ref = fx["polygons"][0], fx["b_a0_a1"][0]           # (6.7,2.2)-(9.3,4.8)
    mine = scenes.rect_halfspaces(poly[:, 0].min(), poly[:, 0].max(), poly[:, 1].min(), poly[:, 1].max())
    mine = mine.reshape(3, 4)
    key = lambda m: sorted(map(tuple, np.round(m.T, 9).tolist()))
    assert key(mine) == key(ref)


def test_polygon_halfspace_representation_matches_all_reference_fixtures_in_row_order():
    """a17: every one of the 16 (polygon -> b, a0, a1) pairs produced by the reference's own function; the ROW ORDER is
    part of the layout (`update_static_constraints` writes b + a0 + a1 in facet order, interface_mpc.py:60-63)."""
    from trajtrack_mpcndqn_rlboost_amd import geometry
    fx = np.load(os.path.join(ROOT, "tests", "golden", "halfspace.npz"))
    assert len(fx["polygons"]) == 16
    for poly, ref in zip(fx["polygons"], fx["b_a0_a1"]):
        b, a0, a1 = geometry.polygon_halfspace_representation(poly)
        mine = np.stack([b, a0, a1])
        assert mine.shape == ref.shape
        assert np.max(np.abs(mine - ref)) <= 1e-12 * max(1.0, np.max(np.abs(ref)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "mpcgpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mpcgpu_[a-z_0-9]+)\s*\(", text)))


def test_c_abi_library_exports_every_declared_symbol():
    path = solver_mod.library_path()
    assert os.path.exists(path), f"{path} missing -- run __graft_entry__.build()"
    lib = ctypes.CDLL(path)
    declared = _declared_symbols()
    assert set(declared) == set(solver_mod.EXPORTS)
    for sym in declared:
        assert hasattr(lib, sym), sym
    lib.mpcgpu_abi_version.restype = ctypes.c_int32
    assert lib.mpcgpu_abi_version() == 8


def test_option_numbers_of_the_ctypes_side_match_the_header():
    """mpcgpu_set_option takes a number: solver.py's OPT_* constants are the header's MPCGPU_OPT_* enumerators, every one of them."""
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "mpcgpu.h")).read(), flags=re.S)
    header = {k: int(v) for k, v in re.findall(r"\bMPCGPU_(OPT_[A-Z_]+)\s*=\s*(\d+)", text)}
    python = {k: v for k, v in vars(solver_mod).items() if k.startswith("OPT_")}
    assert header == python and len(set(header.values())) == len(header) >= 9


def test_variant_builds_export_the_abi_and_their_debug_hooks():
    """Test-only builds (csrc/Makefile `variants`, compiled by __graft_entry__.build()): same C-ABI; the trace build adds
    its two debug entry points, the product library must not carry them."""
    prod = ctypes.CDLL(solver_mod.library_path())
    assert not hasattr(prod, "mpcgpu_debug_set_trace") and not hasattr(prod, "mpcgpu_debug_read_trace")
    for name in ("trace", "lbfgs_lds", "twoloop", "onesite", "linear40", "rowwalk40"):
        path = solver_mod.variant_path(name)
        assert os.path.exists(path), f"{path} missing -- run __graft_entry__.build()"
        lib = ctypes.CDLL(path)
        for sym in _declared_symbols():
            assert hasattr(lib, sym), (name, sym)
        assert hasattr(lib, "mpcgpu_debug_set_trace") == (name == "trace")


def test_c_struct_layout_matches_header():
    """Field order of the ctypes mirror == field order in include/mpcgpu.h."""
    text = open(os.path.join(ROOT, "include", "mpcgpu.h")).read()
    body = text[text.index("typedef struct mpcgpu_config {"):text.index("} mpcgpu_config;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for typ, decl in re.findall(r"\b(int32_t|double)\s+([^;]+);", body):
        names += [n.strip() for n in decl.split(",")]
    assert names == [n for n, _ in solver_mod._CConfig._fields_]
    assert ctypes.sizeof(solver_mod._CConfig) == 8 * 4 + 17 * 8 + 4 * 4 + 8


def test_product_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from trajtrack_mpcndqn_rlboost_amd import BatchSolver, MpcGpuError
    with pytest.raises(MpcGpuError, match="no HIP device|no CPU fallback"):
        BatchSolver(MpcConfig())


def test_create_rejects_unsupported_configs_before_touching_the_gpu():
    from trajtrack_mpcndqn_rlboost_amd import BatchSolver, MpcGpuError
    for kw, pat in ((dict(N_hor=65), "N_hor"), (dict(nstcobs=9), "nstcobs"), (dict(nu=3), "unicycle"),
                    (dict(Ndynobs=0), "Ndynobs"), (dict(solver_lbfgs_memory=0), "lbfgs_mem")):
        with pytest.raises(MpcGpuError, match=pat):
            BatchSolver(MpcConfig(**kw))


def test_no_product_module_imports_the_oracle():
    pkg = os.path.join(ROOT, "trajtrack_mpcndqn_rlboost_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "mpc_oracle" not in src, f


def test_realtime_capacity_search_is_a_bisection_over_fleet_sizes(monkeypatch):
    """tools/closed_loop.realtime_capacity (bench.py: config.closed_loop.realtime_robots_per_gpu): the largest multiple of `step`
    robots whose WORST tick stays within the limit -- with the device loop replaced by a model T(B) = 80 ms + B / 41.5 per ms."""
    import sys
    sys.path.insert(0, ROOT)
    from tools import closed_loop
    calls = []

    def fake(cfg, B, ticks, warmup, n_dyn, warm, order, device=0, **kw):
        calls.append(B)
        worst = 80.0 + B / 41.5
        return {"ms_per_tick_min_max": [10.0, worst], "ms_per_tick": 0.6 * worst, "ms_of_every_tick": [worst] * ticks, "converged_fraction": 0.4}
    monkeypatch.setattr(closed_loop, "device_closed_loop", fake)

    class Cfg:
        ts = 0.2
    r = closed_loop.realtime_capacity(Cfg(), lo=2048, hi=8192, step=512)
    assert r["limit_ms"] == 200.0 and r["robots"] == 4608            # 80 + 4608 / 41.5 = 191 ms; 5120 -> 203 ms
    assert r["robots"] % 512 == 0 and len(calls) <= 6 and calls[0] == 2048 and calls[1] == 8192
    assert set(r["tried"]) == {str(b) for b in calls} and len(r["ms_of_every_tick"]) == 30
    assert closed_loop.realtime_capacity(Cfg(), lo=2048, hi=4096, step=512)["robots"] == 4096      # the whole range fits
    assert closed_loop.realtime_capacity(Cfg(), lo=6144, hi=8192, step=512)["robots"] == 0         # nothing fits
