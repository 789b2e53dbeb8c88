#!/usr/bin/env python3
"""Generate golden vectors for the NMPC cost / constraint / gradient path FROM THE REFERENCE ITSELF.

Run in the build container only (needs /root/reference, which does not exist on the GPU box):

    python tests/golden/make_fixtures.py

How it works: the reference defines the problem symbolically with CasADi + opengen
(/root/reference/src/mpc_traj_tracker/mpc/mpc_generator.py:160-297); neither package is installed.
This script registers two tiny stand-in modules in ``sys.modules`` BEFORE importing the reference:

* ``casadi.casadi``: a float64 torch-backed ``SX`` with CasADi's matrix semantics for exactly the
  operations mpc_generator.py / motion_model.py use (column vectors, column-major linear indexing,
  scalar and column-repeat broadcasting, ``sq(sqrt(x)) -> x`` simplification, CasADi's sub-gradient
  conventions for fmin/fmax);
* ``opengen.opengen``: capturing no-op builder classes (``Problem``, ``Rectangle``, config objects).

Then ``MpcModule(cfg).build(unicycle_model)`` -- the reference's OWN code -- is executed on concrete
numbers, and torch autograd differentiates it.  Outputs (inputs + expected values only, no reference
source) are written next to this file:

    costgrad_N20.npz / costgrad_N40.npz : u, p, c, y -> f, grad f, F1, F2, psi, grad psi, bounds
    problem_meta.json                    : dims, bounds, solver settings captured from the build call
    unicycle_rk4.npz                     : numpy unicycle_model I/O (motion_model.py:142-164)
    halfspace.npz                        : polygon_halfspace_representation I/O (util/utils_geo.py:33-59)

psi follows opengen's published construction (not in /root/reference):
    psi = f + c/2 * dist^2_C(F1 + y/max(c,1)) + c/2 * ||F2||^2.
"""
from __future__ import annotations

import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
torch.set_default_dtype(torch.float64)


# --------------------------------------------------------------------------------------------
# casadi stand-in
# --------------------------------------------------------------------------------------------
class SX:
    """2-D float64 matrix with the subset of CasADi SX semantics the reference uses."""

    _leaves: dict = {}

    def __init__(self, t, sqrt_of=None):
        if not isinstance(t, torch.Tensor):
            t = torch.tensor(t, dtype=torch.float64)
        if t.dim() == 0:
            t = t.reshape(1, 1)
        elif t.dim() == 1:
            t = t.reshape(-1, 1)
        self.t = t
        self._sqrt_of = sqrt_of

    # -- construction ------------------------------------------------------------------
    @staticmethod
    def sym(name, n, m=1):
        st = SX._leaves
        if name == "u":
            assert n * m == st["u"].numel()
            return SX(st["u"].reshape(n, m))
        off = st["off"]
        st["off"] = off + n * m
        return SX(st["p"][off:off + n * m].reshape(n, m))

    @staticmethod
    def ones(n, m=1):
        return SX(torch.ones(n, m))

    @property
    def shape(self):
        return tuple(self.t.shape)

    @property
    def T(self):
        return SX(self.t.t())

    def _flat(self):  # column-major
        return self.t.t().reshape(-1)

    def __getitem__(self, key):
        flat = self._flat()
        if isinstance(key, int):
            return SX(flat[key].reshape(1, 1))
        if isinstance(key, slice):
            return SX(flat[key].reshape(-1, 1))
        raise TypeError(key)

    # -- arithmetic --------------------------------------------------------------------
    @staticmethod
    def _coerce(x):
        return x if isinstance(x, SX) else SX(torch.tensor(float(x)))

    @staticmethod
    def _bcast(a, b):
        a, b = SX._coerce(a).t, SX._coerce(b).t
        if a.shape == b.shape or a.numel() == 1 or b.numel() == 1:
            return a, b
        if a.shape[0] == b.shape[0] and b.shape[1] % a.shape[1] == 0:
            return a.repeat(1, b.shape[1] // a.shape[1]), b
        if a.shape[0] == b.shape[0] and a.shape[1] % b.shape[1] == 0:
            return a, b.repeat(1, a.shape[1] // b.shape[1])
        raise ValueError(f"dimension mismatch {tuple(a.shape)} vs {tuple(b.shape)}")

    def __add__(self, o): a, b = SX._bcast(self, o); return SX(a + b)
    def __radd__(self, o): a, b = SX._bcast(o, self); return SX(a + b)
    def __sub__(self, o): a, b = SX._bcast(self, o); return SX(a - b)
    def __rsub__(self, o): a, b = SX._bcast(o, self); return SX(a - b)
    def __mul__(self, o): a, b = SX._bcast(self, o); return SX(a * b)
    def __rmul__(self, o): a, b = SX._bcast(o, self); return SX(a * b)
    def __truediv__(self, o): a, b = SX._bcast(self, o); return SX(a / b)
    def __rtruediv__(self, o): a, b = SX._bcast(o, self); return SX(a / b)
    def __neg__(self): return SX(-self.t)

    def __pow__(self, e):
        if e == 2:
            if self._sqrt_of is not None:  # CasADi simplifies sq(sqrt(x)) -> x on the fly
                return self._sqrt_of
            return SX(self.t * self.t)
        return SX(self.t ** e)


def _u(f):
    return lambda x: SX(f(SX._coerce(x).t))


def _sqrt(x):
    x = SX._coerce(x)
    return SX(torch.sqrt(x.t), sqrt_of=x)


def _fmax(x, y):  # CasADi: d/dx = (x >= y), d/dy = !(x >= y)
    a, b = SX._bcast(x, y)
    return SX(torch.where(a >= b, a, b))


def _fmin(x, y):  # CasADi: d/dx = (x <= y), d/dy = !(x <= y)
    a, b = SX._bcast(x, y)
    return SX(torch.where(a <= b, a, b))


def _vertcat(*xs):
    return SX(torch.cat([SX._coerce(x).t for x in xs], dim=0))


def _horzcat(*xs):
    return SX(torch.cat([SX._coerce(x).t for x in xs], dim=1))


def _mmin(x):
    flat = x._flat()
    r = flat[0]
    for i in range(1, flat.numel()):  # left fold with fmin: ties keep the accumulated operand
        r = torch.where(r <= flat[i], r, flat[i])
    return SX(r)


def _DM(x):
    return SX(torch.tensor(np.asarray(x, dtype=float)))


cs_mod = types.ModuleType("casadi.casadi")
cs_mod.SX = SX
cs_mod.DM = _DM
cs_mod.cos = _u(torch.cos)
cs_mod.sin = _u(torch.sin)
cs_mod.acos = _u(torch.acos)
cs_mod.exp = _u(torch.exp)
cs_mod.sign = _u(torch.sign)
cs_mod.sqrt = _sqrt
cs_mod.fmax = _fmax
cs_mod.fmin = _fmin
cs_mod.vertcat = _vertcat
cs_mod.vcat = lambda xs: _vertcat(*xs)
cs_mod.horzcat = _horzcat
cs_mod.hcat = lambda xs: _horzcat(*xs)
cs_mod.transpose = lambda x: x.T
cs_mod.mtimes = lambda a, b: SX(SX._coerce(a).t @ SX._coerce(b).t)
cs_mod.dot = lambda a, b: SX((SX._coerce(a).t * SX._coerce(b).t).sum())
cs_mod.sum1 = lambda x: SX(x.t.sum(dim=0, keepdim=True))
cs_mod.sum2 = lambda x: SX(x.t.sum(dim=1, keepdim=True))
cs_mod.mmin = _mmin
cs_mod.norm_2 = lambda x: SX(torch.sqrt((x.t * x.t).sum()))
casadi_pkg = types.ModuleType("casadi")
casadi_pkg.casadi = cs_mod
sys.modules["casadi"] = casadi_pkg
sys.modules["casadi.casadi"] = cs_mod

# --------------------------------------------------------------------------------------------
# opengen stand-in (captures what the reference hands to the builder)
# --------------------------------------------------------------------------------------------
CAPTURE: dict = {}


class _Chain:
    def __init__(self, *a, **k):
        self.calls = {}

    def __getattr__(self, name):
        if name.startswith("with_"):
            def f(*a, **k):
                self.calls[name] = a[0] if len(a) == 1 else a
                return self
            return f
        raise AttributeError(name)


class _Rectangle:
    def __init__(self, xmin, xmax):
        self.xmin, self.xmax = list(xmin), list(xmax)


class _Problem:
    def __init__(self, u, p, cost):
        CAPTURE.update(u=u, p=p, cost=cost)

    def with_constraints(self, c): CAPTURE["U"] = c; return self
    def with_aug_lagrangian_constraints(self, f1, C, Y=None): CAPTURE.update(F1=f1, C=C); return self
    def with_penalty_constraints(self, f2): CAPTURE["F2"] = f2; return self


class _Builder(_Chain):
    def __init__(self, problem, meta, build_cfg, solver_cfg):
        super().__init__()
        CAPTURE.update(meta=meta.calls, build_cfg=build_cfg.calls, solver_cfg=solver_cfg.calls)

    def build(self):
        return None


og = types.ModuleType("opengen.opengen")
og.constraints = types.SimpleNamespace(Rectangle=_Rectangle)
og.builder = types.SimpleNamespace(Problem=_Problem, OpEnOptimizerBuilder=_Builder)
og.config = types.SimpleNamespace(BuildConfiguration=_Chain, OptimizerMeta=_Chain, SolverConfiguration=_Chain)
og_pkg = types.ModuleType("opengen")
og_pkg.opengen = og
og_pkg.constraints, og_pkg.builder, og_pkg.config = og.constraints, og.builder, og.config
sys.modules["opengen"] = og_pkg
sys.modules["opengen.opengen"] = og

sys.path.insert(0, os.path.join(REF, "src"))
from util.mpc_config import Configurator  # noqa: E402  (reference)
from mpc_traj_tracker.mpc.mpc_generator import MpcModule  # noqa: E402  (reference)
from pkg_motion_model.motion_model import unicycle_model  # noqa: E402  (reference)
from util.utils_geo import polygon_halfspace_representation  # noqa: E402  (reference)


# --------------------------------------------------------------------------------------------
def load_cfg(N):
    src = open(os.path.join(REF, "config", "mpc_default.yaml")).read()
    if N != 20:  # mpc_longiter.yaml also says N_hor: 20; N=40 is a synthetic override
        assert "N_hor: 20" in src
        src = src.replace("N_hor: 20", f"N_hor: {N}")
    with tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False) as fh:
        fh.write(src)
        path = fh.name
    cfg = Configurator(path, verbose=False)
    os.unlink(path)
    return cfg


def np_of(cfg):
    N = cfg.N_hor
    return (2 * cfg.ns + cfg.nu) + cfg.nq + (cfg.ns * N + N) + cfg.ns * N * cfg.Nother + \
        cfg.Nstcobs * cfg.nstcobs + cfg.Ndynobs * cfg.ndynobs * N + 2 * N


def evaluate(cfg, u_np, p_np, c, y_np):
    """Run the reference's MpcModule.build on concrete numbers; return dict of float64 arrays."""
    import contextlib, io
    u = torch.tensor(u_np, dtype=torch.float64, requires_grad=True)
    p = torch.tensor(p_np, dtype=torch.float64)
    SX._leaves = dict(u=u, p=p, off=0)
    CAPTURE.clear()
    with contextlib.redirect_stdout(io.StringIO()):
        MpcModule(cfg).build(unicycle_model)
    assert SX._leaves["off"] == p.numel(), (SX._leaves["off"], p.numel())
    cost = CAPTURE["cost"].t.reshape(())
    F1 = CAPTURE["F1"].t.reshape(-1)
    F2 = CAPTURE["F2"].t.reshape(-1)
    lo = torch.tensor(CAPTURE["C"].xmin, dtype=torch.float64)
    hi = torch.tensor(CAPTURE["C"].xmax, dtype=torch.float64)
    (gf,) = torch.autograd.grad(cost, u, retain_graph=True)
    # opengen's psi (published construction; Rectangle.distance_squared)
    z = F1 + torch.tensor(y_np, dtype=torch.float64) / max(c, 1.0)
    dist2 = (torch.clamp(z - hi, min=0.0) ** 2 + torch.clamp(lo - z, min=0.0) ** 2).sum()
    psi = cost + 0.5 * c * dist2 + 0.5 * c * (F2 * F2).sum()
    (gpsi,) = torch.autograd.grad(psi, u)
    return dict(f=cost.item(), grad_f=gf.numpy().copy(), F1=F1.detach().numpy().copy(),
                F2=F2.detach().numpy().copy(), psi=psi.item(), grad_psi=gpsi.numpy().copy())


# --------------------------------------------------------------------------------------------
# input generators (seeded; chosen so every term of the cost is exercised)
# --------------------------------------------------------------------------------------------
def offsets(cfg):
    N = cfg.N_hor
    r0 = 18
    c0 = r0 + 4 * N
    os0 = c0 + 3 * N * cfg.Nother
    od0 = os0 + cfg.Nstcobs * cfg.nstcobs
    qs0 = od0 + cfg.Ndynobs * cfg.ndynobs * N
    qd0 = qs0 + N
    return r0, c0, os0, od0, qs0, qd0, qd0 + N


def rect_halfspaces(x0, x1, y0, y1):
    """b, a0, a1 of an axis-aligned box written directly (inside <=> b - a0 x - a1 y > 0)."""
    return [x1, -x0, y1, -y0], [1.0, -1.0, 0.0, 0.0], [0.0, 0.0, 1.0, -1.0]


def make_case(cfg, rng, kind):
    N = cfg.N_hor
    r0, c0, os0, od0, qs0, qd0, npar = offsets(cfg)
    p = np.zeros(npar)
    x0, y0, th0 = rng.uniform(0, 2), rng.uniform(2, 5), rng.uniform(-0.6, 0.6)
    p[0:3] = [x0, y0, th0]
    p[6:8] = [rng.uniform(0, 1.2), rng.uniform(-0.3, 0.3)]
    # reference polyline: straight then a corner, 0.24 m spacing
    head = th0 + rng.uniform(-0.4, 0.4)
    pts = []
    x, y = x0, y0
    turn_at = rng.integers(3, N)
    for k in range(N):
        if k == turn_at:
            head += rng.uniform(-1.2, 1.2)
        x += 0.24 * np.cos(head); y += 0.24 * np.sin(head)
        pts.append((x, y, head))
    if kind != "dense" and rng.random() < 0.5:  # tail-padded local reference (trajectory_generator.py:223-226)
        cut = rng.integers(N // 2, N)
        pts = pts[:cut] + [pts[cut - 1]] * (N - cut)
    for k, (x, y, h) in enumerate(pts):
        p[r0 + 3 * k:r0 + 3 * k + 3] = [x, y, h]
    p[3:6] = pts[-1]
    p[r0 + 3 * N:r0 + 4 * N] = rng.uniform(0.3, 1.2)
    if kind == "dense":
        p[8:18] = rng.uniform(0.5, 50.0, 10)          # every weight non-zero (incl. rv, rw, qN, qthetaN)
        n_other, n_stc, n_dyn = cfg.Nother, cfg.Nstcobs, cfg.Ndynobs
    else:
        p[8:18] = [0.0, 10.0, 0.0, 0.0, 0.0, 0.0, 0.0, 100.0, 10.0, 20.0]  # yaml defaults via set_work_mode
        n_other, n_stc, n_dyn = rng.integers(0, 3), rng.integers(0, 6), rng.integers(0, 9)
    # robots near the path
    for j in range(n_other):
        k0 = rng.integers(0, N)
        bx, by = pts[k0][0] + rng.uniform(-0.4, 0.4), pts[k0][1] + rng.uniform(-0.4, 0.4)
        vx, vy = rng.uniform(-0.05, 0.05, 2)
        for k in range(N):
            p[c0 + j * 3 * N + 3 * k:c0 + j * 3 * N + 3 * k + 3] = [bx + vx * k, by + vy * k, 0.0]
    # static boxes on / near the path (so that S > 0 happens)
    for o in range(n_stc):
        k0 = rng.integers(0, N)
        cx, cy = pts[k0][0] + rng.uniform(-0.5, 0.5), pts[k0][1] + rng.uniform(-0.5, 0.5)
        hx, hy = rng.uniform(0.3, 1.0, 2)
        b, a0, a1 = rect_halfspaces(cx - hx, cx + hx, cy - hy, cy + hy)
        if kind == "dense":  # rotate the box: general half-planes
            ang = rng.uniform(0, np.pi)
            ca, sa = np.cos(ang), np.sin(ang)
            nrm = [(ca, sa), (-ca, -sa), (-sa, ca), (sa, -ca)]
            ext = [hx, hx, hy, hy]
            b = [ext[e] + nrm[e][0] * cx + nrm[e][1] * cy for e in range(4)]
            a0 = [nrm[e][0] for e in range(4)]
            a1 = [nrm[e][1] for e in range(4)]
        p[os0 + 12 * o:os0 + 12 * o + 12] = b + a0 + a1
    # dynamic ellipses crossing the path
    for i in range(n_dyn):
        k0 = rng.integers(0, N)
        bx, by = pts[k0][0] + rng.uniform(-0.8, 0.8), pts[k0][1] + rng.uniform(-0.8, 0.8)
        vx, vy = rng.uniform(-0.15, 0.15, 2)
        rx, ry = rng.uniform(0.3, 1.6, 2)
        ang = rng.uniform(-np.pi, np.pi)
        for k in range(N):
            alpha = rng.uniform(0.2, 1.0)
            p[od0 + i * 6 * N + 6 * k:od0 + i * 6 * N + 6 * k + 6] = \
                [bx + vx * (k - k0), by + vy * (k - k0), rx, ry, ang, alpha]
    p[qs0:qs0 + N] = 1e3 if kind != "dense" else rng.uniform(1, 100, N)
    p[qd0:qd0 + N] = 1e3 if kind != "dense" else rng.uniform(1, 100, N)
    # decision vector: inside, on, and slightly outside the input box
    v = rng.uniform(-0.6, 1.7, N)
    w = rng.uniform(-0.6, 0.6, N)
    if kind == "zero_u":
        v[:] = 0.0; w[:] = 0.0
    u = np.stack([v, w], axis=1).reshape(-1)
    c = float(rng.choice([0.0, 1.0, 10.0, 50.0, 1250.0]))
    y = rng.uniform(-5, 5, 2 * N) * (rng.random() < 0.7)
    return u, p, c, y


def survey_anchor(cfg):
    """SURVEY.md Appendix C.2 anchor (hand-checked f(u=0) = 16819.2)."""
    N = cfg.N_hor
    r0, c0, os0, od0, qs0, qd0, npar = offsets(cfg)
    p = np.zeros(npar)
    p[0:3] = [0.6, 3.5, 0.0]
    for k in range(N):
        p[r0 + 3 * k:r0 + 3 * k + 3] = [0.6 + 0.24 * (k + 1), 3.5, 0.0]
    p[3:6] = p[r0 + 3 * (N - 1):r0 + 3 * N]
    p[8:18] = [0.0, 10.0, 0.0, 0.0, 0.0, 0.0, 0.0, 100.0, 10.0, 20.0]
    p[r0 + 3 * N:r0 + 4 * N] = 1.2
    p[os0:os0 + 12] = [3, -2, 4, -3, 1, -1, 0, 0, 0, 0, 1, -1]
    for k in range(N):
        p[od0 + 6 * k:od0 + 6 * k + 6] = [4.0 - 0.1 * k, 3.6, 0.8, 0.6, 0.3, 1.0]
    p[qs0:qs0 + N] = 1e3
    p[qd0:qd0 + N] = 1e3
    return p


def gen_costgrad(N, n_cases, seed):
    cfg = load_cfg(N)
    rng = np.random.default_rng(seed)
    U, P, Cc, Yy, out = [], [], [], [], []
    cases = []
    if N == 20:
        p = survey_anchor(cfg)
        cases.append((np.zeros(2 * N), p, 0.0, np.zeros(2 * N)))
        r = np.random.default_rng(0)
        v = r.uniform(-0.2, 1.2, (N, 1)); w = r.uniform(-0.3, 0.3, (N, 1))
        cases.append((np.concatenate([v, w], axis=1).reshape(-1), p, 10.0, np.zeros(2 * N)))
    kinds = ["dense", "default", "default", "zero_u"]
    while len(cases) < n_cases:
        cases.append(make_case(cfg, rng, kinds[len(cases) % len(kinds)]))
    for (u, p, c, y) in cases:
        assert p.size == np_of(cfg)
        r = evaluate(cfg, u, p, c, y)
        assert np.all(np.isfinite(r["grad_psi"])), "non-finite gradient in fixture"
        U.append(u); P.append(p); Cc.append(c); Yy.append(y); out.append(r)
    meta = dict(
        N=N, np=int(np_of(cfg)), n1=len(CAPTURE["C"].xmin), n2=int(out[0]["F2"].size),
        U_lo=CAPTURE["U"].xmin, U_hi=CAPTURE["U"].xmax, C_lo=CAPTURE["C"].xmin, C_hi=CAPTURE["C"].xmax,
        solver_cfg={k: (v if not isinstance(v, tuple) else list(v)) for k, v in CAPTURE["solver_cfg"].items()},
        optimizer_name=CAPTURE["meta"].get("with_optimizer_name"),
        yaml={k: getattr(cfg, k) for k in ("ts", "N_hor", "nu", "ns", "nq", "Nother", "Nstcobs", "nstcobs",
                                          "Ndynobs", "ndynobs", "vehicle_width", "social_margin",
                                          "lin_vel_min", "lin_vel_max", "ang_vel_max", "lin_acc_min",
                                          "lin_acc_max", "ang_acc_max")})
    np.savez_compressed(
        os.path.join(HERE, f"costgrad_N{N}.npz"),
        u=np.array(U), p=np.array(P), c=np.array(Cc), y=np.array(Yy),
        f=np.array([o["f"] for o in out]), grad_f=np.array([o["grad_f"] for o in out]),
        F1=np.array([o["F1"] for o in out]), F2=np.array([o["F2"] for o in out]),
        psi=np.array([o["psi"] for o in out]), grad_psi=np.array([o["grad_psi"] for o in out]))
    return meta, out


def gen_unicycle(seed=7):
    rng = np.random.default_rng(seed)
    S = rng.uniform(-3, 3, (64, 3)); A = np.stack([rng.uniform(-0.5, 1.5, 64), rng.uniform(-0.5, 0.5, 64)], 1)
    out = np.array([unicycle_model(S[i], A[i], 0.2) for i in range(64)])
    np.savez_compressed(os.path.join(HERE, "unicycle_rk4.npz"), state=S, action=A, ts=0.2, next_state=out)


def gen_halfspace(seed=11):
    rng = np.random.default_rng(seed)
    polys, outs = [], []
    polys.append(np.array([(6.7, 2.2), (9.3, 2.2), (9.3, 4.8), (6.7, 4.8)]))  # SURVEY.md 8c probe
    for _ in range(15):
        cx, cy = rng.uniform(0, 10, 2); hx, hy = rng.uniform(0.3, 2, 2); ang = rng.uniform(0, np.pi)
        R = np.array([[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]])
        corners = np.array([(-hx, -hy), (hx, -hy), (hx, hy), (-hx, hy)]) @ R.T + [cx, cy]
        polys.append(corners)
    for poly in polys:
        b, a0, a1 = polygon_halfspace_representation(np.array(poly))
        outs.append(np.array([b, a0, a1]))
    np.savez_compressed(os.path.join(HERE, "halfspace.npz"), polygons=np.array(polys), b_a0_a1=np.array(outs))


if __name__ == "__main__":
    metas = {}
    m20, out20 = gen_costgrad(20, 48, seed=2024)
    metas["N20"] = m20
    print("N=20: anchor f(u=0) =", out20[0]["f"], " f(rand) =", out20[1]["f"])
    assert abs(out20[0]["f"] - 16819.2) < 1e-6
    m40, _ = gen_costgrad(40, 12, seed=2025)
    metas["N40"] = m40
    gen_unicycle()
    gen_halfspace()
    with open(os.path.join(HERE, "problem_meta.json"), "w") as fh:
        json.dump(metas, fh, indent=1)
    print("fixtures written to", HERE)
