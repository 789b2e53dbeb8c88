"""ctypes binding of ``libmpcgpu.so`` (C-ABI: ``include/mpcgpu.h``).

``BatchSolver.solve`` is the batched counterpart of the reference's ``solver.run(p, initial_guess)``
(``src/mpc_traj_tracker/trajectory_generator.py:318``).  The library is loaded from this package directory
(built in-tree by ``csrc/Makefile``); if it is missing, or no HIP device is usable, construction raises --
there is deliberately no CPU path.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass
from typing import Optional

import numpy as np

from .config import MpcConfig

_PKG = os.path.dirname(os.path.abspath(__file__))
_LIB = os.environ.get("MPCGPU_LIB", os.path.join(_PKG, "libmpcgpu.so"))  # override: kernel A/B experiments

STATUS_NAMES = ("Converged", "NotConvergedIterations", "NotConvergedOutOfTime", "NotFiniteComputation",
                "ShapeExceeded")
ABI_VERSION = 8
_STREAM_OWN = C.c_void_p(-1)      # MPCGPU_STREAM_OWN: the handle's own non-blocking stream
OPT_LINESEARCH_FALLBACK = 1       # MPCGPU_OPT_LINESEARCH_FALLBACK
OPT_PAIRING = 2                   # MPCGPU_OPT_PAIRING
OPT_TEAM_BATCH = 3                # MPCGPU_OPT_TEAM_BATCH
OPT_ORDER = 4                     # MPCGPU_OPT_ORDER
OPT_LINEAR_TABLES = 5             # MPCGPU_OPT_LINEAR_TABLES
OPT_TAIL_PROMOTION = 6            # MPCGPU_OPT_TAIL_PROMOTION
OPT_TAIL_POLL = 7                 # MPCGPU_OPT_TAIL_POLL
OPT_TAIL_WAVES = 8                # MPCGPU_OPT_TAIL_WAVES
OPT_TAIL_CONCURRENT = 9           # MPCGPU_OPT_TAIL_CONCURRENT
OPT_PENALTY_STALL = 10            # MPCGPU_OPT_PENALTY_STALL
OPT_TAIL_GRADUAL = 11             # MPCGPU_OPT_TAIL_GRADUAL


def _stream_arg(stream):
    """``None`` -> the handle's own stream; an int is a raw hipStream_t used as given (0 = HIP's null stream =
    torch's default stream, ordered with the torch work around the call)."""
    return _STREAM_OWN if stream is None else C.c_void_p(int(stream))


class MpcGpuError(RuntimeError):
    """Any failure reported through the C-ABI (the reference's callers catch RuntimeError)."""


class _CConfig(C.Structure):
    _fields_ = [
        ("N", C.c_int32), ("nu", C.c_int32), ("ns", C.c_int32), ("Nother", C.c_int32), ("Nstcobs", C.c_int32),
        ("nstcobs", C.c_int32), ("Ndynobs", C.c_int32), ("ndynobs", C.c_int32),
        ("ts", C.c_double),
        ("lin_vel_min", C.c_double), ("lin_vel_max", C.c_double), ("ang_vel_max", C.c_double),
        ("lin_acc_min", C.c_double), ("lin_acc_max", C.c_double), ("ang_acc_max", C.c_double),
        ("vehicle_width", C.c_double), ("social_margin", C.c_double), ("fleet_weight", C.c_double),
        ("tol", C.c_double), ("delta_tol", C.c_double), ("init_tol", C.c_double), ("init_penalty", C.c_double),
        ("penalty_update", C.c_double), ("tol_update", C.c_double), ("suff_decrease", C.c_double),
        ("max_inner", C.c_int32), ("max_outer", C.c_int32), ("lbfgs_mem", C.c_int32), ("device", C.c_int32),
        ("max_duration_us", C.c_double),
    ]


EXPORTS = ("mpcgpu_abi_version", "mpcgpu_create", "mpcgpu_destroy", "mpcgpu_last_error", "mpcgpu_num_params",
           "mpcgpu_solve_batch", "mpcgpu_solve_batch_dev", "mpcgpu_cost_grad_batch", "mpcgpu_last_timing",
           "mpcgpu_last_eval_counts", "mpcgpu_last_shape", "mpcgpu_last_waves_per_simd", "mpcgpu_reserve_shape",
           "mpcgpu_set_option", "mpcgpu_last_problems_per_wavefront", "mpcgpu_last_ordered", "mpcgpu_last_tail_promotion", "mpcgpu_last_tail_timeouts", "mpcgpu_last_tail_timing", "mpcgpu_last_latency_kernel", "mpcgpu_reserve_batch", "mpcgpu_last_table_kind", "mpcgpu_tracker_window_dev",
           "mpcgpu_tracker_step_dev", "mpcgpu_rl_reference_dev", "mpcgpu_hint_switch_dev", "mpcgpu_debug_read_workspace",
           "mpcgpu_workspace_stride", "mpcgpu_workspace_record", "mpcgpu_debug_prep", "mpcgpu_debug_tracker_assemble",
           "mpcgpu_debug_lbfgs_direction")


def library_path() -> str:
    return _LIB


def build_library(force: bool = False) -> str:
    """Compile libmpcgpu.so for gfx950 with hipcc (recipe: csrc/Makefile)."""
    cmd = ["make", "-C", os.path.join(_PKG, "csrc"), "-s"]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd)
    return _LIB


class CTracker(C.Structure):
    """``mpcgpu_tracker`` (include/mpcgpu.h): device view of the tracker state of B robots."""
    _fields_ = [("B", C.c_int32), ("ref_cap", C.c_int32), ("action_steps", C.c_int32), ("_pad", C.c_int32),
                ("states", C.c_void_p), ("goals", C.c_void_p), ("last_actions", C.c_void_p), ("ref", C.c_void_p),
                ("ref_len", C.c_void_p), ("idx_ref", C.c_void_p), ("stc", C.c_void_p), ("dyn", C.c_void_p),
                ("other", C.c_void_p), ("pred_states", C.c_void_p), ("active", C.c_void_p),
                ("tuning", C.c_double * 10), ("base_speed", C.c_double), ("low_speed", C.c_double),
                ("stc_weight", C.c_double), ("dyn_weight", C.c_double)]


_libs = {}


def variant_path(name: str) -> str:
    """Path of a test-only variant build (csrc/Makefile `variants`): 'trace', 'lbfgs_lds', 'twoloop', 'onesite', 'linear40'."""
    return os.path.join(_PKG, "variants", f"libmpcgpu_{name}.so")


def load_library(path: Optional[str] = None):
    """ctypes handle of libmpcgpu.so (or of another build of it given by ``path``: the tests load the decision-trace
    and L-BFGS-in-LDS variants next to the product library)."""
    _LIB = os.path.abspath(path) if path else globals()["_LIB"]
    if _LIB in _libs:
        return _libs[_LIB]
    # PyTorch's ROCm wheel bundles its own libamdhip64 / libhsa-runtime64 (same SONAME as the system's).  If torch is
    # imported first, our library binds to that already-loaded runtime and the process has ONE HIP runtime (torch
    # tensors' pointers and stream handles are then first-class here).  The other order puts two runtimes in the
    # process and torch no longer sees the GPU -- so torch, when installed, always goes first.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(_LIB):
        raise MpcGpuError(f"{_LIB} not found: build it with `make -C {os.path.join(_PKG, 'csrc')}` "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    L = C.CDLL(_LIB)
    dp, ip, vp = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.c_void_p
    L.mpcgpu_abi_version.restype = C.c_int32
    L.mpcgpu_create.argtypes = [C.POINTER(_CConfig), C.POINTER(vp)]
    L.mpcgpu_create.restype = C.c_int32
    L.mpcgpu_destroy.argtypes = [vp]
    L.mpcgpu_destroy.restype = None
    L.mpcgpu_last_error.argtypes = [vp]
    L.mpcgpu_last_error.restype = C.c_char_p
    L.mpcgpu_num_params.argtypes = [vp]
    L.mpcgpu_num_params.restype = C.c_int32
    L.mpcgpu_solve_batch.argtypes = [vp, C.c_int32, dp, dp, dp, dp, dp, dp, ip, ip, ip, dp, dp, dp, dp]
    L.mpcgpu_solve_batch.restype = C.c_int32
    L.mpcgpu_solve_batch_dev.argtypes = [vp, C.c_int32] + [vp] * 13 + [vp]
    L.mpcgpu_solve_batch_dev.restype = C.c_int32
    L.mpcgpu_cost_grad_batch.argtypes = [vp, C.c_int32, dp, dp, dp, dp, dp, dp, dp, dp]
    L.mpcgpu_cost_grad_batch.restype = C.c_int32
    L.mpcgpu_last_timing.argtypes = [vp, dp, dp]
    L.mpcgpu_last_timing.restype = C.c_int32
    L.mpcgpu_last_eval_counts.argtypes = [vp, C.c_int32, ip, ip, vp]
    L.mpcgpu_last_eval_counts.restype = C.c_int32
    L.mpcgpu_last_shape.argtypes = [vp, ip, ip, ip, ip]
    L.mpcgpu_last_shape.restype = C.c_int32
    L.mpcgpu_last_waves_per_simd.argtypes = [vp]
    L.mpcgpu_last_waves_per_simd.restype = C.c_int32
    L.mpcgpu_last_latency_kernel.argtypes = [vp]
    L.mpcgpu_last_latency_kernel.restype = C.c_int32
    L.mpcgpu_last_problems_per_wavefront.argtypes = [vp]
    L.mpcgpu_last_problems_per_wavefront.restype = C.c_int32
    L.mpcgpu_last_ordered.argtypes = [vp]
    L.mpcgpu_last_ordered.restype = C.c_int32
    if hasattr(L, "mpcgpu_last_tail_timeouts"):
        L.mpcgpu_last_tail_timeouts.argtypes = [vp, ip, vp]
        L.mpcgpu_last_tail_timeouts.restype = C.c_int32
    if hasattr(L, "mpcgpu_last_tail_promotion"):   # (absent only in an old build loaded for an A/B run, see MPCGPU_ALLOW_ABI)
        L.mpcgpu_last_tail_promotion.argtypes = [vp, ip, vp]
        L.mpcgpu_last_tail_promotion.restype = C.c_int32
        L.mpcgpu_last_tail_timing.argtypes = [vp, dp, dp]
        L.mpcgpu_last_tail_timing.restype = C.c_int32
    L.mpcgpu_reserve_shape.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
    L.mpcgpu_reserve_shape.restype = C.c_int32
    tp = C.POINTER(CTracker)
    L.mpcgpu_tracker_window_dev.argtypes = [vp, tp, vp, vp]
    L.mpcgpu_tracker_window_dev.restype = C.c_int32
    L.mpcgpu_tracker_step_dev.argtypes = [vp, tp] + [vp] * 8 + [vp]
    L.mpcgpu_tracker_step_dev.restype = C.c_int32
    L.mpcgpu_rl_reference_dev.argtypes = [vp, C.c_int32, vp, C.c_int32, vp, C.c_double, C.c_int32, C.c_double, dp, vp, vp]
    L.mpcgpu_rl_reference_dev.restype = C.c_int32
    L.mpcgpu_hint_switch_dev.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp, vp, vp, vp, vp, C.c_int32, vp,
                                         C.c_double, C.c_double, C.c_double, vp, vp, vp, vp]
    L.mpcgpu_hint_switch_dev.restype = C.c_int32
    L.mpcgpu_debug_read_workspace.argtypes = [vp, C.c_int32, dp]
    L.mpcgpu_debug_read_workspace.restype = C.c_int32
    L.mpcgpu_workspace_stride.argtypes = [vp]
    L.mpcgpu_workspace_stride.restype = C.c_int32
    L.mpcgpu_workspace_record.argtypes = [vp]
    L.mpcgpu_workspace_record.restype = C.c_int32
    L.mpcgpu_debug_prep.argtypes = [vp, C.c_int32, dp]
    L.mpcgpu_debug_prep.restype = C.c_int32
    L.mpcgpu_debug_tracker_assemble.argtypes = [vp, tp, vp]
    L.mpcgpu_debug_tracker_assemble.restype = C.c_int32
    L.mpcgpu_debug_lbfgs_direction.argtypes = [vp, C.c_int32, C.c_int32, dp, dp, dp, dp, ip]
    L.mpcgpu_debug_lbfgs_direction.restype = C.c_int32
    L.mpcgpu_last_table_kind.argtypes = [vp]
    L.mpcgpu_last_table_kind.restype = C.c_int32
    L.mpcgpu_reserve_batch.argtypes = [vp, C.c_int32]
    L.mpcgpu_reserve_batch.restype = C.c_int32
    L.mpcgpu_set_option.argtypes = [vp, C.c_int32, C.c_double]
    L.mpcgpu_set_option.restype = C.c_int32
    if L.mpcgpu_abi_version() != ABI_VERSION and not os.environ.get("MPCGPU_ALLOW_ABI"):   # (A/B runs against an older build)
        raise MpcGpuError(f"{_LIB} has ABI version {L.mpcgpu_abi_version()}, this binding needs {ABI_VERSION}: rebuild it")
    if hasattr(L, "mpcgpu_debug_set_trace"):     # -DMPC_TRACE builds (tests)
        L.mpcgpu_debug_set_trace.argtypes = [vp, C.c_int32]
        L.mpcgpu_debug_set_trace.restype = C.c_int32
        L.mpcgpu_debug_read_trace.argtypes = [vp, C.c_int32, dp]
        L.mpcgpu_debug_read_trace.restype = C.c_int32
    _libs[_LIB] = L
    return L


@dataclass
class BatchResult:
    """Per-problem fields of the OpEn solution object (``solution, cost, exit_status, solve_time_ms`` are the
    ones the reference reads at trajectory_generator.py:320-323)."""
    solution: np.ndarray        # [B, 2N]
    cost: np.ndarray            # [B]
    status: np.ndarray          # [B] int32 codes, see STATUS_NAMES
    num_inner_iterations: np.ndarray
    num_outer_iterations: np.ndarray
    last_problem_norm_fpr: np.ndarray
    f2_norm: np.ndarray
    lagrange_multipliers: np.ndarray  # [B, 2N]
    solve_time_ms: np.ndarray

    @property
    def exit_status(self):
        return [STATUS_NAMES[s] for s in self.status]


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_int32))


class BatchSolver:
    """One handle of libmpcgpu.so = one generated solver of the reference, for batches of problems."""

    def __init__(self, config: Optional[MpcConfig] = None, device: int = 0, library: Optional[str] = None,
                 pairing: Optional[int] = None, latency_batch: Optional[int] = None, order: Optional[str] = None,
                 linear_tables: Optional[bool] = None, tail_promotion: Optional[int] = None):
        """``pairing``: problems per wavefront of the solve kernel -- None = the library's rule (the faster layout: one),
        1 or 2 to force a layout (MPCGPU_OPT_PAIRING; 2 exists for N_hor = 20; env MPCGPU_PAIRING overrides None).
        ``latency_batch``: largest batch solved by the latency kernel (MPCGPU_OPT_TEAM_BATCH; None = the library's rule,
        0 = never; env MPCGPU_TEAM_BATCH overrides None).
        ``order``: in which order the throughput kernel starts the problems of a large batch (MPCGPU_OPT_ORDER): None /
        "longest_first" = by the evaluation counts of this handle's previous call of the same batch size (the library's
        default: robot i of this tick is robot i of the last one), "as_given" = workgroup g solves problem g (env MPCGPU_ORDER
        overrides None).  Results do not depend on it (bitwise).
        ``linear_tables``: False = never use the linear centre tables (MPCGPU_OPT_LINEAR_TABLES = 0; env MPCGPU_LINEAR_TABLES=0
        overrides None; an experiment that only the variant build libmpcgpu_linear40.so carries).  Results do not depend on it.
        ``tail_promotion``: how many problems at the end of a throughput launch move to the latency kernel at the start of their
        next inner problem (MPCGPU_OPT_TAIL_PROMOTION): None / -1 = the library's rule (2 per compute unit), 0 = off (env
        MPCGPU_TAIL_PROMOTION overrides None).  Results do not depend on it (bitwise).  Where those problems run -- beside the
        draining launch on a stream of the handle's own (default) or behind it: ``set_tail_concurrent`` / env MPCGPU_TAIL_CONCURRENT."""
        self.config = config if config is not None else MpcConfig()
        self._L = load_library(library)
        self._h = C.c_void_p()
        d = self.config.solver_dict(device)
        self._cfg = _CConfig(**{name: d[name] for name, _ in _CConfig._fields_})
        rc = self._L.mpcgpu_create(C.byref(self._cfg), C.byref(self._h))
        if rc != 0:
            msg = self._L.mpcgpu_last_error(None).decode()
            self._h = C.c_void_p()
            raise MpcGpuError(f"mpcgpu_create failed ({rc}): {msg}")
        self.device = device
        self.N = int(self.config.N_hor)
        self.n = 2 * self.N
        self.np = int(self._L.mpcgpu_num_params(self._h))
        assert self.np == self.config.num_params
        fb = getattr(self.config, "solver_linesearch_fallback", "last_trial")
        if fb not in ("last_trial", "half_step"):
            raise MpcGpuError(f"solver_linesearch_fallback must be 'last_trial' or 'half_step', got {fb!r}")
        self._check(self._L.mpcgpu_set_option(self._h, OPT_LINESEARCH_FALLBACK, 1.0 if fb == "half_step" else 0.0),
                    "mpcgpu_set_option")
        stall = getattr(self.config, "solver_penalty_stall", "either")
        if stall not in ("either", "both"):
            raise MpcGpuError(f"solver_penalty_stall must be 'either' or 'both', got {stall!r}")
        self._check(self._L.mpcgpu_set_option(self._h, OPT_PENALTY_STALL, 1.0 if stall == "both" else 0.0),
                    "mpcgpu_set_option")
        if pairing is None and os.environ.get("MPCGPU_PAIRING"):
            pairing = int(os.environ["MPCGPU_PAIRING"])
        if latency_batch is None and os.environ.get("MPCGPU_TEAM_BATCH"):
            latency_batch = int(os.environ["MPCGPU_TEAM_BATCH"])
        if latency_batch is not None:
            self._check(self._L.mpcgpu_set_option(self._h, OPT_TEAM_BATCH, float(latency_batch)), "mpcgpu_set_option")
        if pairing is not None:
            if pairing not in (1, 2):
                raise MpcGpuError(f"pairing must be 1 or 2 problems per wavefront, got {pairing!r}")
            self._check(self._L.mpcgpu_set_option(self._h, OPT_PAIRING, float(pairing - 1)), "mpcgpu_set_option")
        if order is None and os.environ.get("MPCGPU_ORDER"):
            order = os.environ["MPCGPU_ORDER"]
        if order is not None:
            self.set_order(order)
        if linear_tables is None and os.environ.get("MPCGPU_LINEAR_TABLES"):
            linear_tables = os.environ["MPCGPU_LINEAR_TABLES"] != "0"
        if linear_tables is not None:
            self._check(self._L.mpcgpu_set_option(self._h, OPT_LINEAR_TABLES, 1.0 if linear_tables else 0.0), "mpcgpu_set_option")

        if tail_promotion is None and os.environ.get("MPCGPU_TAIL_PROMOTION"):
            tail_promotion = int(os.environ["MPCGPU_TAIL_PROMOTION"])
        if tail_promotion is not None:
            self.set_tail_promotion(tail_promotion)
        if os.environ.get("MPCGPU_TAIL_CONCURRENT"):
            self.set_tail_concurrent(int(os.environ["MPCGPU_TAIL_CONCURRENT"]) != 0)
        if os.environ.get("MPCGPU_TAIL_GRADUAL"):
            self.set_tail_gradual(int(os.environ["MPCGPU_TAIL_GRADUAL"]))

    def set_tail_promotion(self, problems: int, poll_steps: Optional[int] = None, waves: Optional[int] = None):
        """MPCGPU_OPT_TAIL_PROMOTION (-1 automatic, 0 off, K problems) and, for the A/B build that can leave inside an inner
        problem, MPCGPU_OPT_TAIL_POLL."""
        self._check(self._L.mpcgpu_set_option(self._h, OPT_TAIL_PROMOTION, float(problems)), "mpcgpu_set_option")
        if poll_steps is not None:
            self._check(self._L.mpcgpu_set_option(self._h, OPT_TAIL_POLL, float(poll_steps)), "mpcgpu_set_option")
        if waves is not None:
            self._check(self._L.mpcgpu_set_option(self._h, OPT_TAIL_WAVES, float(waves)), "mpcgpu_set_option")

    def set_tail_concurrent(self, on: bool):
        """MPCGPU_OPT_TAIL_CONCURRENT: the continuation of the tail promotion runs on a stream of the handle's own while the
        throughput launch drains (True) or as the launch behind it (False).  Results do not depend on it (bitwise)."""
        self._check(self._L.mpcgpu_set_option(self._h, OPT_TAIL_CONCURRENT, 1.0 if on else 0.0), "mpcgpu_set_option")

    def set_tail_gradual(self, finished_per_promoted: int):
        """MPCGPU_OPT_TAIL_GRADUAL: with the concurrent continuation, a problem may also leave the throughput launch (at the start of an inner
        problem, once every problem of the launch has begun) while (promoted + 1) * finished_per_promoted <= finished; 0 = off.  Results
        do not depend on it (bitwise)."""
        self._check(self._L.mpcgpu_set_option(self._h, OPT_TAIL_GRADUAL, float(finished_per_promoted)), "mpcgpu_set_option")

    def last_tail_promotion(self, stream: Optional[int] = None):
        """(capacity of the continuation launch of the last solve call, problems that actually moved to the latency kernel)."""
        n = C.c_int32()
        cap = int(self._L.mpcgpu_last_tail_promotion(self._h, C.byref(n), _stream_arg(stream)))
        return cap, int(n.value)

    def last_tail_timeouts(self, stream: Optional[int] = None):
        """(the last solve ran its continuation beside the draining launch, number of its bounded waits that ended by their wall-clock
        limit: 0 in a healthy run -- a starved side stream costs time, the sweep launch finishes what is left, same results)."""
        n = C.c_int32()
        conc = int(self._L.mpcgpu_last_tail_timeouts(self._h, C.byref(n), _stream_arg(stream)))
        if conc < 0:
            self._check(conc, "mpcgpu_last_tail_timeouts")
        return bool(conc), int(n.value)

    def set_order(self, order: str):
        if order not in ("as_given", "longest_first"):
            raise MpcGpuError(f"order must be 'as_given' or 'longest_first', got {order!r}")
        self._check(self._L.mpcgpu_set_option(self._h, OPT_ORDER, 1.0 if order == "longest_first" else 0.0), "mpcgpu_set_option")

    # -- lifetime ---------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.mpcgpu_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, what: str):
        if rc != 0:
            raise MpcGpuError(f"{what} failed ({rc}): {self._L.mpcgpu_last_error(self._h).decode()}")

    # -- host-pointer API -------------------------------------------------------------------------
    def solve(self, p, initial_guess=None, initial_lagrange_multipliers=None, initial_penalty=None) -> BatchResult:
        p = np.ascontiguousarray(p, dtype=np.float64)
        if p.ndim == 1:
            p = p[None]
        if p.ndim != 2 or p.shape[1] != self.np:
            raise MpcGpuError(f"3003 -> wrong number of parameters: got {p.shape}, expected [B, {self.np}]")
        B = p.shape[0]

        def opt(a, shape, what, code):
            if a is None:
                return None
            a = np.ascontiguousarray(a, dtype=np.float64)
            if a.shape != shape:
                raise MpcGpuError(f"{code} -> {what} has incompatible dimensions: got {a.shape}, expected {shape}")
            return a
        u0 = opt(initial_guess, (B, self.n), "initial guess", 1600)
        y0 = opt(initial_lagrange_multipliers, (B, self.n), "Lagrange multipliers", 1700)
        c0 = opt(initial_penalty, (B,), "initial penalty", 1800)
        u = np.empty((B, self.n)); cost = np.empty(B); status = np.empty(B, np.int32)
        inner = np.empty(B, np.int32); outer = np.empty(B, np.int32)
        fpr = np.empty(B); f2 = np.empty(B); y = np.empty((B, self.n)); ms = np.empty(B)
        rc = self._L.mpcgpu_solve_batch(self._h, B, _dp(p), _dp(u0), _dp(y0), _dp(c0), _dp(u), _dp(cost),
                                        _ip(status), _ip(inner), _ip(outer), _dp(fpr), _dp(f2), _dp(y), _dp(ms))
        self._check(rc, "mpcgpu_solve_batch")
        return BatchResult(u, cost, status, inner, outer, fpr, f2, y, ms)

    def solve_into(self, p: np.ndarray, out: dict, initial_guess=None):
        """``mpcgpu_solve_batch`` (HOST pointers: what the reference's plugin boundary hands over, trajectory_generator.py:318) on
        caller-owned float64 / int32 numpy arrays, nothing allocated here: ``p [B, np]``, ``out`` with ``u [B, 2N]``, ``cost [B]``,
        ``status [B]`` (int32) and optionally ``inner_it``, ``outer_it`` (int32).  The arrays may be views of page-locked memory
        (``torch.empty(..., pin_memory=True).numpy()``).  Synchronous: results are in place on return."""
        B = int(p.shape[0])
        if p.dtype != np.float64 or not p.flags.c_contiguous or p.shape != (B, self.np):
            raise MpcGpuError(f"p must be a C-contiguous float64 [B, {self.np}] array, got {p.dtype} {p.shape}")
        for name, dt, shape in (("u", np.float64, (B, self.n)), ("cost", np.float64, (B,)), ("status", np.int32, (B,))):
            a = out[name]
            if a.dtype != dt or a.shape != shape or not a.flags.c_contiguous:
                raise MpcGpuError(f"out[{name!r}] must be a C-contiguous {np.dtype(dt).name} {shape} array")
        u0 = None if initial_guess is None else np.ascontiguousarray(initial_guess, dtype=np.float64)
        rc = self._L.mpcgpu_solve_batch(self._h, B, _dp(p), _dp(u0), None, None, _dp(out["u"]), _dp(out["cost"]), _ip(out["status"]),
                                        _ip(out.get("inner_it")), _ip(out.get("outer_it")), None, None, None, None)
        self._check(rc, "mpcgpu_solve_batch")

    def cost_grad(self, u, p, c=None, y=None):
        """Test hook: psi, f, grad psi, F1, F2 of every problem (GPU evaluation of the generated functions)."""
        p = np.ascontiguousarray(p, dtype=np.float64)
        u = np.ascontiguousarray(u, dtype=np.float64)
        if p.ndim == 1:
            p, u = p[None], u[None]
        B = p.shape[0]
        if p.shape != (B, self.np) or u.shape != (B, self.n):
            raise MpcGpuError(f"bad shapes p{p.shape} u{u.shape}")
        xi = np.zeros((B, 1 + self.n))
        if c is not None:
            xi[:, 0] = c
        if y is not None:
            xi[:, 1:] = y
        psi = np.empty(B); f = np.empty(B); grad = np.empty((B, self.n)); F1 = np.empty((B, self.n))
        F2 = np.empty((B, int(self.config.Ndynobs)))
        rc = self._L.mpcgpu_cost_grad_batch(self._h, B, _dp(u), _dp(xi), _dp(p), _dp(psi), _dp(f), _dp(grad),
                                            _dp(F1), _dp(F2))
        self._check(rc, "mpcgpu_cost_grad_batch")
        return dict(psi=psi, f=f, grad=grad, F1=F1, F2=F2)

    # -- device-pointer API (torch tensors on this solver's device) ---------------------------------
    def solve_device(self, p, out: dict, initial_guess=None, initial_lagrange_multipliers=None,
                     initial_penalty=None, stream: Optional[int] = None):
        """Enqueue a batch whose inputs/outputs are float64/int32 CUDA tensors (HBM-resident).

        ``out`` must hold preallocated tensors ``u [B,2N] f64, cost [B] f64, status [B] i32`` and may hold
        ``inner_it, outer_it (i32), fpr, f2norm, ms (f64 [B]), y (f64 [B,2N])``.  ``stream`` is a raw
        hipStream_t used as given -- pass ``torch.cuda.current_stream().cuda_stream`` (0 for torch's default
        stream = HIP's null stream) and the solve is ordered with the torch kernels that produce ``p`` and consume
        ``out``; ``None`` = the handle's own non-blocking stream (no ordering with torch: synchronise yourself).
        Returns after the solve kernel has been enqueued (without blocking when ``reserve_shape`` was called).
        """
        B = int(p.shape[0])
        if tuple(p.shape) != (B, self.np) or not p.is_contiguous():
            raise MpcGpuError(f"p must be a contiguous [B, {self.np}] tensor, got {tuple(p.shape)}")

        def ptr(t):
            return None if t is None else C.c_void_p(t.data_ptr())
        rc = self._L.mpcgpu_solve_batch_dev(
            self._h, B, ptr(p), ptr(initial_guess), ptr(initial_lagrange_multipliers), ptr(initial_penalty),
            ptr(out["u"]), ptr(out["cost"]), ptr(out["status"]), ptr(out.get("inner_it")),
            ptr(out.get("outer_it")), ptr(out.get("fpr")), ptr(out.get("f2norm")), ptr(out.get("y")),
            ptr(out.get("ms")), _stream_arg(stream))
        self._check(rc, "mpcgpu_solve_batch_dev")

    def reserve_shape(self, max_static: Optional[int] = None, max_fleet: Optional[int] = None,
                      max_dyn: Optional[int] = None, var_shape: bool = True, axis_aligned: bool = False):
        """Promise upper bounds on the active rows of the following ``solve_device`` batches (``None`` = the
        configured maximum): the launch then needs no count read-back -- it never blocks and can be captured into
        a hipGraph (after ``reserve_batch``).  ``var_shape=False``: every dynamic row keeps (rx, ry, angle, alpha) over
        the horizon (compact tables); with ``axis_aligned=True`` on top: every row has angle 0, what the reference's own
        prediction feeder produces (src/main.py:77-85).  Problems that break a promise come back with status 4
        (``ShapeExceeded``)."""
        c = self.config
        if axis_aligned and var_shape:
            raise MpcGpuError("axis_aligned=True needs var_shape=False")
        self._check(self._L.mpcgpu_reserve_shape(
            self._h, int(c.Nstcobs if max_static is None else max_static),
            int(c.Nother if max_fleet is None else max_fleet), int(c.Ndynobs if max_dyn is None else max_dyn),
            1 if var_shape else (2 if axis_aligned else 0)), "mpcgpu_reserve_shape")

    def reserve_batch(self, B: int):
        """Size the library-owned device buffers for batches of up to ``B`` problems now (needed before a
        ``solve_device`` call is captured into a hipGraph: nothing may be allocated inside a capture)."""
        self._check(self._L.mpcgpu_reserve_batch(self._h, int(B)), "mpcgpu_reserve_batch")

    # -- batched tracker harness on the device (include/mpcgpu.h: mpcgpu_tracker_*; device_tracker.DeviceTracker) ------------
    def tracker_window(self, view: "CTracker", refs_out, stream: Optional[int] = None):
        self._check(self._L.mpcgpu_tracker_window_dev(self._h, C.byref(view), C.c_void_p(refs_out.data_ptr()), _stream_arg(stream)),
                    "mpcgpu_tracker_window_dev")

    def tracker_step(self, view: "CTracker", refs, out: dict, initial_guess=None, stream: Optional[int] = None):
        """One tick of all robots: termination test, assembly (compact record, no padded vector), solve, taken / predicted
        states.  ``out``: ``u [B,2N], cost [B], status [B]`` and optionally ``inner_it, outer_it, actions [B,2]``."""
        def ptr(t):
            return None if t is None else C.c_void_p(t.data_ptr())
        self._check(self._L.mpcgpu_tracker_step_dev(self._h, C.byref(view), ptr(refs), ptr(initial_guess), ptr(out["u"]),
                                                    ptr(out["cost"]), ptr(out["status"]), ptr(out.get("inner_it")),
                                                    ptr(out.get("outer_it")), ptr(out.get("actions")), _stream_arg(stream)),
                    "mpcgpu_tracker_step_dev")

    def rl_reference(self, agent, action, ts: float, steps: int, ref_speed: float, limits, rl_ref, stream: Optional[int] = None):
        lim = (C.c_double * 8)(*[float(v) for v in limits])
        self._check(self._L.mpcgpu_rl_reference_dev(self._h, int(agent.shape[0]), C.c_void_p(agent.data_ptr()), int(agent.stride(0)),
                                                    C.c_void_p(action.data_ptr()), float(ts), int(steps), float(ref_speed), lim,
                                                    C.c_void_p(rl_ref.data_ptr()), _stream_arg(stream)), "mpcgpu_rl_reference_dev")

    def hint_switch(self, polygons, valid, states, original, rl_ref, live, distances, switch_on, detach_cnt, chosen,
                    stream: Optional[int] = None):
        B, N = int(original.shape[0]), int(original.shape[1])
        O, V = int(polygons.shape[1]), int(polygons.shape[2])
        self._check(self._L.mpcgpu_hint_switch_dev(
            self._h, B, N, O, V, C.c_void_p(polygons.data_ptr()), C.c_void_p(valid.data_ptr()), C.c_void_p(states.data_ptr()),
            C.c_void_p(original.data_ptr()), C.c_void_p(rl_ref.data_ptr()), int(rl_ref.shape[1]),
            None if live is None else C.c_void_p(live.data_ptr()), float(distances[0]), float(distances[1]), float(distances[2]),
            C.c_void_p(switch_on.data_ptr()), C.c_void_p(detach_cnt.data_ptr()), C.c_void_p(chosen.data_ptr()), _stream_arg(stream)),
            "mpcgpu_hint_switch_dev")

    def debug_workspace(self, B: int):
        """Test hook: (records [B, stride], length of the compact record at the start of each)."""
        stride, rec = int(self._L.mpcgpu_workspace_stride(self._h)), int(self._L.mpcgpu_workspace_record(self._h))
        out = np.empty((B, stride))
        self._check(self._L.mpcgpu_debug_read_workspace(self._h, B, _dp(out)), "mpcgpu_debug_read_workspace")
        return out, rec

    def debug_prep(self, p):
        p = np.ascontiguousarray(p, dtype=np.float64)
        self._check(self._L.mpcgpu_debug_prep(self._h, int(p.shape[0]), _dp(p)), "mpcgpu_debug_prep")

    def debug_tracker_assemble(self, view: "CTracker", refs):
        self._check(self._L.mpcgpu_debug_tracker_assemble(self._h, C.byref(view), C.c_void_p(refs.data_ptr())),
                    "mpcgpu_debug_tracker_assemble")

    def debug_lbfgs_direction(self, U, R):
        """Test hook: U, R [B, m + 1, 2N] (iterates, gamma*fpr) -> (d_gram [B, 2N], d_twoloop [B, 2N], pairs [B, 2])."""
        U = np.ascontiguousarray(U, dtype=np.float64); R = np.ascontiguousarray(R, dtype=np.float64)
        B, m1 = U.shape[0], U.shape[1]
        if U.shape != (B, m1, self.n) or R.shape != U.shape or m1 < 2:
            raise MpcGpuError(f"bad shapes U{U.shape} R{R.shape}")
        dg = np.empty((B, self.n)); dt = np.empty((B, self.n)); pairs = np.empty((B, 2), np.int32)
        self._check(self._L.mpcgpu_debug_lbfgs_direction(self._h, B, m1 - 1, _dp(U), _dp(R), _dp(dg), _dp(dt), _ip(pairs)),
                    "mpcgpu_debug_lbfgs_direction")
        return dg, dt, pairs

    def release_shape(self):
        self._check(self._L.mpcgpu_reserve_shape(self._h, -1, -1, -1, 0), "mpcgpu_reserve_shape")

    # -- decision trace (only with a -DMPC_TRACE build of the library; tests) ------------------------
    def set_trace(self, cap: int):
        if not hasattr(self._L, "mpcgpu_debug_set_trace"):
            raise MpcGpuError("this libmpcgpu.so was not built with -DMPC_TRACE")
        self._check(self._L.mpcgpu_debug_set_trace(self._h, int(cap)), "mpcgpu_debug_set_trace")
        self._trace_cap = int(cap)

    def read_trace(self, B: int) -> np.ndarray:
        """[B, cap, 12] records of the last solve (fields: oracle.TRACE_FIELDS); NaN rows were not written."""
        out = np.empty((B, self._trace_cap, 12))
        self._check(self._L.mpcgpu_debug_read_trace(self._h, B, _dp(out)), "mpcgpu_debug_read_trace")
        return out

    def last_timing(self):
        """prep_ms, solve_ms (HIP events around the compaction and around everything the solve enqueued) and the split of solve_ms
        at the tail promotion: main_ms (throughput kernel) + tail_ms (continuation launch of the latency kernel, 0 without one)."""
        a, b = C.c_double(), C.c_double()
        self._check(self._L.mpcgpu_last_timing(self._h, C.byref(a), C.byref(b)), "mpcgpu_last_timing")
        out = dict(prep_ms=a.value, solve_ms=b.value, main_ms=b.value, tail_ms=0.0)
        if hasattr(self._L, "mpcgpu_last_tail_timing"):
            m, t = C.c_double(), C.c_double()
            self._check(self._L.mpcgpu_last_tail_timing(self._h, C.byref(m), C.byref(t)), "mpcgpu_last_tail_timing")
            out["main_ms"], out["tail_ms"] = m.value, t.value
        return out

    def last_eval_counts(self, B: int, stream: Optional[int] = None):
        """(psi evaluations, of which with gradient) per problem of the last solve of B problems (``stream``: the
        one that solve was enqueued on, same convention as ``solve_device``)."""
        n_psi = np.empty(B, np.int32); n_grad = np.empty(B, np.int32)
        self._check(self._L.mpcgpu_last_eval_counts(self._h, B, _ip(n_psi), _ip(n_grad), _stream_arg(stream)),
                    "mpcgpu_last_eval_counts")
        return n_psi, n_grad

    def last_shape(self):
        v = [C.c_int32() for _ in range(4)]
        self._check(self._L.mpcgpu_last_shape(self._h, *[C.byref(x) for x in v]), "mpcgpu_last_shape")
        return dict(max_static=v[0].value, max_fleet=v[1].value, max_dyn=v[2].value, lds_bytes=v[3].value,
                    waves_per_simd=int(self._L.mpcgpu_last_waves_per_simd(self._h)),
                    problems_per_wavefront=int(self._L.mpcgpu_last_problems_per_wavefront(self._h)),
                    ordered=bool(self._L.mpcgpu_last_ordered(self._h)),
                    latency_kernel=bool(self._L.mpcgpu_last_latency_kernel(self._h)),
                    shape_const=int(self._L.mpcgpu_last_table_kind(self._h)) != 1,
                    axis_aligned=int(self._L.mpcgpu_last_table_kind(self._h)) >= 2,
                    linear=int(self._L.mpcgpu_last_table_kind(self._h)) == 3)
