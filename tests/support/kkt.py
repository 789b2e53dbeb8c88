"""TEST INFRASTRUCTURE: a solution-level check that does not go through any PANOC / ALM restatement.

A point u* returned by the solver (GPU kernel or CPU oracle) is examined as a candidate local minimiser of the
reference's CONSTRAINED problem (/root/reference/src/mpc_traj_tracker/mpc/mpc_generator.py:248-272):

    minimise f(u)   s.t.   u in U (input box),   F1(u) in C (acceleration box),   F2(u) = 0 (hard obstacle constraints)

with f, grad f, F1 and F2 taken from ``oracle.cost_grad`` -- the functions pinned bit for bit by the fixtures that
``tests/golden/make_fixtures.py`` generates from the reference's own CasADi graph -- and scipy as the only optimiser:

  1. feasibility of u* to delta (the solver's own delta_tolerance, 1e-4);
  2. SLSQP started at u* on the same problem (hard constraints in their natural inequality form, `inside_ellipses <= 0`
     and `min over the edges of the half-plane value <= 0`, mpc_generator.py:38-54, evaluated on a rollout made of
     ``oracle.unicycle_rk4`` steps) must not move it by more than 1e-3 nor find an f that is lower by more than 1e-6
     relative;
  3. the projected-gradient residual of the Lagrangian || u* - Proj_U(u* - grad_u L(u*, y*, mu)) ||_inf, with the
     multipliers y* the solver returns for F1 and the best non-negative multipliers mu of the ACTIVE hard constraints
     (non-negative least squares), must be small.

F2 in equality form has a vanishing Jacobian on the feasible set, which no SQP method can work with; the inequality
form above describes the same feasible set, and ``hard_constraints`` is checked against F2 of the oracle at every
call (F2_i = S + sum_k max(0, inside_ellipse_ik), S = sum of the products of squared hinges)."""
from __future__ import annotations

import numpy as np
from scipy.optimize import LinearConstraint, NonlinearConstraint, minimize, nnls

import oracle


def rollout(cfg, p, u):
    """Positions after each of the N steps (RK4 unicycle, motion_model.py:142-164 through the oracle)."""
    N = int(cfg.N_hor)
    s = np.array(p[0:3], dtype=float)
    X = np.empty((N, 3))
    for k in range(N):
        s = oracle.unicycle_rk4(s, u[2 * k:2 * k + 2], float(cfg.ts))
        X[k] = s
    return X


def hard_constraints(cfg, p, u, with_parts=False):
    """g(u) <= 0: one entry per (active dynamic row, step) -- the hard ellipse indicator -- and per (active static
    polygon, step) -- the smallest half-plane value (positive = strictly inside)."""
    N = int(cfg.N_hor)
    off = cfg.offsets()
    X = rollout(cfg, p, u)
    g = []
    dyn = p[off["od"]:off["qstc"]].reshape(cfg.Ndynobs, N, 6)
    ih_rows = []
    for i in range(cfg.Ndynobs):
        if not np.any(dyn[i]):
            # zero padding = degenerate ellipses at the origin (semi-axes 1e-6): cannot be entered
            ih_rows.append(np.zeros(N))
            continue
        cx, cy, rx, ry, ang = (dyn[i, :, j] for j in range(5))
        ex, ey = X[:, 0] - cx, X[:, 1] - cy
        ih = 1.0 - (ex * np.cos(ang) + ey * np.sin(ang)) ** 2 / (rx + 1e-6) ** 2 \
                 - (ex * np.sin(ang) - ey * np.cos(ang)) ** 2 / (ry + 1e-6) ** 2
        ih_rows.append(ih)
        g.append(ih)
    stc = p[off["os"]:off["od"]].reshape(cfg.Nstcobs, 12)
    S = 0.0
    for o in range(cfg.Nstcobs):
        if not np.any(stc[o]):
            continue
        b, a0, a1 = stc[o, 0:4], stc[o, 4:8], stc[o, 8:12]
        h = b[None, :] - X[:, 0:1] * a0[None, :] - X[:, 1:2] * a1[None, :]      # [N, 4]
        g.append(h.min(axis=1))
        S += float(np.prod(np.maximum(h, 0.0) ** 2, axis=1).sum())
    gv = np.concatenate(g) if g else np.zeros(0)
    if with_parts:
        F2 = np.array([S + np.maximum(r, 0.0).sum() for r in ih_rows])
        n_dyn_entries = N * sum(1 for i in range(cfg.Ndynobs) if np.any(dyn[i]))
        return gv, F2, n_dyn_entries
    return gv


def f1_matrix(ocfg, p):
    """F1(u) = A u + b (acceleration mapping, linear): A column by column from the oracle."""
    n = 2 * ocfg.N
    b = oracle.cost_grad(ocfg, np.zeros(n), p)["F1"]
    A = np.empty((n, n))
    for j in range(n):
        e = np.zeros(n); e[j] = 1.0
        A[:, j] = oracle.cost_grad(ocfg, e, p)["F1"] - b
    return A, b


def check_solution(cfg, ocfg, p, u, y, run_scipy=True, active_tol=1e-3):
    """All quantities of the module docstring for one (p, u*, y*)."""
    N, n = int(cfg.N_hor), 2 * int(cfg.N_hor)
    lo = np.tile([cfg.lin_vel_min, -cfg.ang_vel_max], N)
    hi = np.tile([cfg.lin_vel_max, cfg.ang_vel_max], N)
    clo = np.r_[np.full(N, cfg.lin_acc_min), np.full(N, -cfg.ang_acc_max)]
    chi = np.r_[np.full(N, cfg.lin_acc_max), np.full(N, cfg.ang_acc_max)]
    o = oracle.cost_grad(ocfg, u, p)
    f0, gf, F1, F2 = o["f"], o["grad"], o["F1"], o["F2"]
    g0, F2_restated, n_dyn_entries = hard_constraints(cfg, p, u, with_parts=True)
    # the inequality form describes the oracle's F2 (pinned by the reference-generated fixtures)
    assert np.allclose(F2_restated, F2, rtol=1e-9, atol=1e-12), (F2_restated, F2)
    out = dict(f=f0)
    out["infeas_U"] = float(max(0.0, (lo - u).max(), (u - hi).max()))
    out["infeas_C"] = float(max(0.0, (clo - F1).max(), (F1 - chi).max()))
    out["infeas_F2"] = float(np.abs(F2).max()) if F2.size else 0.0
    # ---- Lagrangian residual: grad f + A' y + sum_j mu_j grad g_j with mu >= 0 on the active hard constraints
    A, _ = f1_matrix(ocfg, p)
    gL = gf + A.T @ y
    act = np.where(g0 > -active_tol)[0]
    out["n_active_hard"] = int(act.size)
    # dynamic ellipses first, then static polygons (order of `hard_constraints`).  The static constraint the solver sees is
    # the PRODUCT of squared hinges: F2 <= delta admits a penetration of centimetres near a polygon edge (flat constraint),
    # which the inequality form used here does not -- `g_static_max` lets the callers tell those answers apart.
    out["n_active_dyn"] = int((act < n_dyn_entries).sum())
    out["g_static_max"] = float(g0[n_dyn_entries:].max()) if g0.size > n_dyn_entries else -np.inf
    if act.size:
        J = np.empty((act.size, n))
        h = 1e-6
        for j in range(n):
            e = np.zeros(n); e[j] = h
            J[:, j] = (hard_constraints(cfg, p, u + e)[act] - hard_constraints(cfg, p, u - e)[act]) / (2 * h)
        free = (u > lo + 1e-9) & (u < hi - 1e-9)              # clamped coordinates are absorbed by the box multipliers
        mu, _ = nnls(J[:, free].T, -gL[free]) if free.any() else (np.zeros(act.size), 0.0)
        gL = gL + J.T @ mu
        out["mu_max"] = float(mu.max())
        out["mu_dyn_max"] = float(mu[act < n_dyn_entries].max()) if (act < n_dyn_entries).any() else 0.0
    out["pg_residual"] = float(np.abs(u - np.clip(u - gL, lo, hi)).max())
    if not run_scipy:
        return out
    # ---- scipy from u*
    cons = [LinearConstraint(A, clo - (F1 - A @ u), chi - (F1 - A @ u))]
    if g0.size:
        cons.append(NonlinearConstraint(lambda x: hard_constraints(cfg, p, x), -np.inf, 0.0))
    res = minimize(lambda x: oracle.cost_grad(ocfg, x, p)["f"], u, jac=lambda x: oracle.cost_grad(ocfg, x, p)["grad"],
                   bounds=list(zip(lo, hi)), constraints=cons, method="SLSQP",
                   options=dict(maxiter=200, ftol=1e-12))
    out["scipy_status"] = int(res.status)
    out["scipy_move"] = float(np.abs(res.x - u).max())
    gs = hard_constraints(cfg, p, res.x)
    out["scipy_feasible"] = bool((gs.max() if gs.size else -1.0) <= 1e-6)
    out["scipy_f_gain_rel"] = float((f0 - res.fun) / max(abs(f0), 1e-12))     # > 0: scipy found a lower f
    return out


def scipy_from_cold_start(cfg, ocfg, p, maxiter=500):
    """An INDEPENDENT solve of the reference's constrained problem: SLSQP on the reference-pinned f / grad f, the acceleration box as
    a linear constraint and the hard constraints in their inequality form, started where the reference's call sites start the solver
    (initial_guess=None -> u = 0, src/interface_mpc.py:82).  Nothing of the PANOC / ALM restatement is involved.  Returns
    dict(x, ok, f, n_active_hard)."""
    N = int(cfg.N_hor)
    lo = np.tile([cfg.lin_vel_min, -cfg.ang_vel_max], N)
    hi = np.tile([cfg.lin_vel_max, cfg.ang_vel_max], N)
    clo = np.r_[np.full(N, cfg.lin_acc_min), np.full(N, -cfg.ang_acc_max)]
    chi = np.r_[np.full(N, cfg.lin_acc_max), np.full(N, cfg.ang_acc_max)]
    A, b0 = f1_matrix(ocfg, p)
    cons = [LinearConstraint(A, clo - b0, chi - b0)]
    if hard_constraints(cfg, p, np.zeros(2 * N)).size:
        cons.append(NonlinearConstraint(lambda x: hard_constraints(cfg, p, x), -np.inf, 0.0))
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = minimize(lambda x: oracle.cost_grad(ocfg, x, p)["f"], np.zeros(2 * N), jac=lambda x: oracle.cost_grad(ocfg, x, p)["grad"],
                       bounds=list(zip(lo, hi)), constraints=cons, method="SLSQP", options=dict(maxiter=maxiter, ftol=1e-13))
    g = hard_constraints(cfg, p, res.x)
    return dict(x=res.x, ok=res.status == 0, f=float(res.fun), n_active_hard=int((g > -1e-3).sum()) if g.size else 0)
