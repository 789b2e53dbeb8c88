"""CPU ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes binding of ``oracle/mpc_oracle.c`` (plain-C double-precision restatement of the reference's
NMPC problem -- /root/reference/src/mpc_traj_tracker/mpc/mpc_generator.py:160-272 -- and of the
published PANOC + ALM/PM algorithm that the reference's generated OpEn solver runs).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package.  The product package ``trajtrack_mpcndqn_rlboost_amd`` never does.

Parity status: cost/gradient/constraints pinned by ``tests/golden`` fixtures generated from the
reference's own source; the solver iteration is "parity unpinned" (no reference golden vectors, OpEn
not buildable here).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libmpc_oracle.so")
_OVERRIDE = os.environ.get("MPC_ORACLE_LIB")   # another build of the same source (tests: the address/UB-sanitizer build)
if _OVERRIDE:
    _LIB_PATH = _OVERRIDE

STATUS_NAMES = ("Converged", "NotConvergedIterations", "NotConvergedOutOfTime")


class OracleConfig(C.Structure):
    _fields_ = [
        ("N", C.c_int32), ("Nother", C.c_int32), ("Nstcobs", C.c_int32), ("nstcobs", C.c_int32),
        ("Ndynobs", C.c_int32), ("ndynobs", C.c_int32),
        ("ts", C.c_double),
        ("lin_vel_min", C.c_double), ("lin_vel_max", C.c_double), ("ang_vel_max", C.c_double),
        ("lin_acc_min", C.c_double), ("lin_acc_max", C.c_double), ("ang_acc_max", C.c_double),
        ("vehicle_width", C.c_double), ("social_margin", C.c_double), ("fleet_weight", C.c_double),
        ("tol", C.c_double), ("delta_tol", C.c_double), ("init_tol", C.c_double),
        ("init_penalty", C.c_double), ("penalty_update", C.c_double), ("tol_update", C.c_double),
        ("suff_decrease", C.c_double),
        ("max_inner", C.c_int32), ("max_outer", C.c_int32), ("lbfgs_mem", C.c_int32), ("ls_fallback", C.c_int32),
        ("lbfgs_gram", C.c_int32), ("stall_rule", C.c_int32),
        ("max_duration_us", C.c_double),
    ]

    @classmethod
    def from_dict(cls, d: dict) -> "OracleConfig":
        cfg = cls()
        for name, _ in cls._fields_:
            setattr(cfg, name, d.get(name, 0) if name in ("ls_fallback", "lbfgs_gram", "stall_rule") else d[name])
        return cfg


class OracleResult(C.Structure):
    _fields_ = [
        ("cost", C.c_double), ("fpr", C.c_double), ("f2_norm", C.c_double), ("delta_y_norm", C.c_double),
        ("penalty", C.c_double), ("solve_time_ms", C.c_double),
        ("status", C.c_int32), ("outer_iters", C.c_int32), ("inner_iters", C.c_int32),
        ("n_cost_evals", C.c_int32), ("n_grad_evals", C.c_int32), ("_pad", C.c_int32),
        ("lbfgs_dev", C.c_double),
    ]


RESULT_DTYPE = np.dtype([
    ("cost", "f8"), ("fpr", "f8"), ("f2_norm", "f8"), ("delta_y_norm", "f8"), ("penalty", "f8"),
    ("solve_time_ms", "f8"), ("status", "i4"), ("outer_iters", "i4"), ("inner_iters", "i4"),
    ("n_cost_evals", "i4"), ("n_grad_evals", "i4"), ("_pad", "i4"), ("lbfgs_dev", "f8")])
assert RESULT_DTYPE.itemsize == C.sizeof(OracleResult)


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (recipe: oracle/Makefile)."""
    src = os.path.join(_HERE, "mpc_oracle.c")
    if _OVERRIDE:
        return _LIB_PATH
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        dp = C.POINTER(C.c_double)
        L.mpc_oracle_np.argtypes = [C.POINTER(OracleConfig)]
        L.mpc_oracle_np.restype = C.c_int32
        L.mpc_oracle_unicycle_rk4.argtypes = [dp, dp, C.c_double, dp]
        L.mpc_oracle_cost_grad.argtypes = [C.POINTER(OracleConfig), dp, C.c_double, dp, dp, dp, dp, dp, dp, dp]
        L.mpc_oracle_solve.argtypes = [C.POINTER(OracleConfig), dp, dp, dp, C.c_double, dp, dp,
                                       C.POINTER(OracleResult)]
        L.mpc_oracle_solve.restype = C.c_int32
        L.mpc_oracle_solve_trace.argtypes = [C.POINTER(OracleConfig), dp, dp, dp, C.c_double, dp, dp,
                                             C.POINTER(OracleResult), dp, C.c_int32, C.POINTER(C.c_int32)]
        L.mpc_oracle_solve_trace.restype = C.c_int32
        L.mpc_oracle_solve_batch.argtypes = [C.POINTER(OracleConfig), C.c_int32, dp, dp, dp, dp, dp, dp,
                                             C.c_void_p, C.c_int32]
        L.mpc_oracle_solve_batch.restype = C.c_int32
        _lib = L
    return _lib


def _dp(a: Optional[np.ndarray]):
    if a is None:
        return None
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        assert a.shape == tuple(shape), (a.shape, shape)
    return a


def num_params(cfg: OracleConfig) -> int:
    return int(lib().mpc_oracle_np(C.byref(cfg)))


def unicycle_rk4(state, action, ts: float) -> np.ndarray:
    s = _f64(state, (3,)); a = _f64(action, (2,)); out = np.empty(3)
    lib().mpc_oracle_unicycle_rk4(_dp(s), _dp(a), float(ts), _dp(out))
    return out


def cost_grad(cfg: OracleConfig, u, p, c: float = 0.0, y=None):
    """Returns dict(f, psi, grad[2N], F1[2N], F2[Ndynobs]) at one point."""
    n = 2 * cfg.N
    u = _f64(u, (n,)); p = _f64(p, (num_params(cfg),))
    y = None if y is None else _f64(y, (n,))
    f = C.c_double(); psi = C.c_double()
    grad = np.empty(n); F1 = np.empty(n); F2 = np.empty(cfg.Ndynobs)
    lib().mpc_oracle_cost_grad(C.byref(cfg), _dp(u), float(c), _dp(y), _dp(p), C.byref(f), C.byref(psi),
                               _dp(grad), _dp(F1), _dp(F2))
    return dict(f=f.value, psi=psi.value, grad=grad, F1=F1, F2=F2)


def solve_batch(cfg: OracleConfig, p, u0=None, y0=None, c0=None, nthreads: int = 0):
    """B independent solves. Returns (u[B,2N], y[B,2N], results structured array, threads_used)."""
    p = _f64(p)
    if p.ndim == 1:
        p = p[None]
    B, n = p.shape[0], 2 * cfg.N
    assert p.shape[1] == num_params(cfg), (p.shape, num_params(cfg))
    u0 = None if u0 is None else _f64(u0, (B, n))
    y0 = None if y0 is None else _f64(y0, (B, n))
    c0 = None if c0 is None else _f64(c0, (B,))
    u = np.empty((B, n)); y = np.empty((B, n))
    res = np.zeros(B, dtype=RESULT_DTYPE)
    used = lib().mpc_oracle_solve_batch(C.byref(cfg), B, _dp(p), _dp(u0), _dp(y0), _dp(c0), _dp(u), _dp(y),
                                        res.ctypes.data_as(C.c_void_p), int(nthreads))
    return u, y, res, int(used)


def solve(cfg: OracleConfig, p, u0=None, y0=None, c0: float = 0.0):
    u, y, res, _ = solve_batch(cfg, np.asarray(p)[None], None if u0 is None else np.asarray(u0)[None],
                               None if y0 is None else np.asarray(y0)[None],
                               None if not c0 else np.array([c0]), nthreads=1)
    return u[0], y[0], res[0]


TRACE_FIELDS = ("outer", "step", "c", "L", "gamma", "nfpr", "psi_u", "n_lip", "lbfgs_pairs", "n_ls", "tau", "psi_next")


def solve_trace(cfg: OracleConfig, p, u0=None, y0=None, c0: float = 0.0, cap: int = 200):
    """One solve with its decision trace: (u, result record, trace[min(steps, cap), 12], steps).  Fields: TRACE_FIELDS."""
    n = 2 * cfg.N
    p = _f64(p, (num_params(cfg),))
    u0 = None if u0 is None else _f64(u0, (n,))
    y0 = None if y0 is None else _f64(y0, (n,))
    u = np.empty(n); y = np.empty(n)
    res = np.zeros(1, dtype=RESULT_DTYPE)
    tr = np.full((cap, len(TRACE_FIELDS)), np.nan)
    steps = C.c_int32()
    rc = lib().mpc_oracle_solve_trace(C.byref(cfg), _dp(p), _dp(u0), _dp(y0), float(c0), _dp(u), _dp(y),
                                      C.cast(res.ctypes.data, C.POINTER(OracleResult)), _dp(tr), cap, C.byref(steps))
    assert rc == 0, rc
    return u, res[0], tr[:min(steps.value, cap)], steps.value
