"""CPU ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Second, independent restatement of the solver ITERATION (PANOC + L-BFGS + ALM/PM) in numpy, written from the
algorithm statement in DESIGN.md section 3 rather than from ``mpc_oracle.c``.  It calls the C oracle only for
psi / grad psi / F1 / F2 evaluations.  ``tests/test_oracle_golden.py`` runs both on small iteration budgets: the C
oracle must follow this restatement step for step, which guards the oracle itself against coding slips (the
published algorithm is the only other anchor: the reference's generated OpEn solver cannot be built here).
Pure-Python loops: use for a handful of problems and a few dozen iterations only.
"""
from __future__ import annotations

import numpy as np

from . import OracleConfig, cost_grad

GAMMA_L = 0.95


class _Lbfgs:
    def __init__(self, mem):
        self.mem = mem
        self.reset()

    def reset(self):
        self.S, self.Y, self.first = [], [], True

    def update(self, g, x):
        if self.first:
            self.first, self.old_x, self.old_g = False, x.copy(), g.copy()
            return
        s, y = x - self.old_x, g - self.old_g
        ys, ss = float(s @ y), float(s @ s)
        if ss <= np.finfo(float).tiny or ys <= 1e-10:
            return
        if not (ys / ss > 1e-8 * np.linalg.norm(g)):
            return
        self.old_x, self.old_g = x.copy(), g.copy()
        self.S.insert(0, s); self.Y.insert(0, y)
        del self.S[self.mem:], self.Y[self.mem:]
        self.h0 = ys / float(y @ y)

    def apply(self, q):
        q = q.copy()
        if not self.S:
            return q
        alphas = []
        for s, y in zip(self.S, self.Y):
            a = float(s @ q) / float(s @ y)
            alphas.append(a)
            q -= a * y
        q *= self.h0
        for s, y, a in reversed(list(zip(self.S, self.Y, alphas))):
            b = float(y @ q) / float(s @ y)
            q += (a - b) * s
        return q


def solve(cfg: OracleConfig, p, u0=None, y0=None, c0=None):
    """Returns dict(u, y, cost, status, inner_iters, outer_iters, penalty)."""
    N = cfg.N
    n = 2 * N
    lo = np.tile([cfg.lin_vel_min, -cfg.ang_vel_max], N)
    hi = np.tile([cfg.lin_vel_max, cfg.ang_vel_max], N)
    c_lo = np.r_[np.full(N, cfg.lin_acc_min), np.full(N, -cfg.ang_acc_max)]
    c_hi = np.r_[np.full(N, cfg.lin_acc_max), np.full(N, cfg.ang_acc_max)]
    u = np.zeros(n) if u0 is None else np.array(u0, dtype=float)
    y = np.zeros(n) if y0 is None else np.array(y0, dtype=float)
    c = cfg.init_penalty if not c0 else float(c0)
    akkt_tol, eps = cfg.init_tol, np.finfo(float).eps
    lb = _Lbfgs(cfg.lbfgs_mem)
    dy_prev = f2_prev = 0.0
    inner_total, status, outer = 0, 0, 0
    y_plus = y.copy()

    def ev(x):
        r = cost_grad(cfg, x, p, c, y)
        return r["psi"], r["grad"]

    for outer in range(1, cfg.max_outer + 1):
        y = np.clip(y, -1e12, 1e12)
        # ---------------- PANOC on psi(.; c, y)
        lb.reset()
        cost, g = ev(u)
        h = np.where(1e-6 * u > 1e-12, 1e-6 * u, 1e-12)
        L = np.linalg.norm(ev(u + h)[1] - g) / np.linalg.norm(h)
        gamma = GAMMA_L / max(L, 1e-10)
        u_half = np.clip(u - gamma * g, lo, hi)
        g_prev = np.zeros(n)
        it = num_iter = 0
        cont = True

        def step():
            nonlocal u, g, g_prev, cost, L, gamma, u_half, it
            if it >= 1:
                g_prev = g.copy()
            r = u - u_half
            nr = np.linalg.norm(r)
            if nr < cfg.tol and np.linalg.norm(r / gamma + g - g_prev) < akkt_tol:
                return False, nr
            cost_half = cost_grad(cfg, u_half, p, c, y)["psi"]
            k = 0
            while cost_half > cost + 1e-6 * abs(cost) - g @ r + GAMMA_L / (2 * gamma) * nr ** 2 and k < 10 and L < 1e9:
                lb.reset()
                L *= 2; gamma /= 2
                u_half = np.clip(u - gamma * g, lo, hi)
                cost_half = cost_grad(cfg, u_half, p, c, y)["psi"]
                r = u - u_half; nr = np.linalg.norm(r); k += 1
            sigma = (1 - GAMMA_L) / (4 * gamma)
            lb.update(r, u)
            if it == 0:
                u = u_half.copy()
                cost, g = ev(u)
                u_half = np.clip(u - gamma * g, lo, hi)
            else:
                d = lb.apply(r)
                gs = u - gamma * g
                rhs = cost - 0.5 * gamma * (g @ g) + 0.5 * np.sum((gs - u_half) ** 2) / gamma - sigma * nr ** 2
                tau, nls = 1.0, 0
                while True:
                    up = u - (1 - tau) * r - tau * d
                    cost, g = ev(up)
                    gs = up - gamma * g
                    u_half = np.clip(gs, lo, hi)
                    lhs = cost - 0.5 * gamma * (g @ g) + 0.5 * np.sum((gs - u_half) ** 2) / gamma
                    if not (lhs > rhs and nls < 10):
                        break
                    tau /= 2; nls += 1
                u = up
            it += 1
            return True, nr

        flag, nfpr = step()
        while flag and cont:
            num_iter += 1
            cont = num_iter < cfg.max_inner
            flag, nfpr = step()
        u = u_half.copy()
        inner_status = 0 if cont else 1
        inner_total += num_iter
        # ---------------- multipliers, infeasibilities, exit test, penalty update
        r = cost_grad(cfg, u, p, c, y)
        F1, F2 = r["F1"], r["F2"]
        y_plus = y + c * (F1 - np.clip(F1 + y / c, c_lo, c_hi))
        dy, f2 = np.linalg.norm(y_plus - y), np.linalg.norm(F2)
        if (outer > 1 and dy <= c * cfg.delta_tol + eps) and f2 <= cfg.delta_tol + eps and akkt_tol <= cfg.tol + eps:
            status = inner_status
            break
        # penalty stall criterion: the penalty is kept in the first outer iteration and when either (stall_rule 0: the published
        # engine as recalled) or both (stall_rule 1: SURVEY.md Appendix B) infeasibilities shrank by suff_decrease
        alm_shrank = dy <= cfg.suff_decrease * dy_prev + eps
        pm_shrank = f2 <= cfg.suff_decrease * f2_prev + eps
        stalled = (alm_shrank and pm_shrank) if cfg.stall_rule == 1 else (alm_shrank or pm_shrank)
        if outer > 1 and not stalled:
            c *= cfg.penalty_update
        akkt_tol = max(akkt_tol * cfg.tol_update, cfg.tol)
        dy_prev, f2_prev, y = dy, f2, y_plus.copy()
    if outer == cfg.max_outer:
        status = 1
    return dict(u=u, y=y_plus, cost=cost_grad(cfg, u, p, 0.0, y)["f"], status=status, inner_iters=inner_total,
                outer_iters=outer, penalty=c)
