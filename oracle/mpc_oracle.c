/*
 * mpc_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).  See mpc_oracle.h.
 *
 * Serial, readable, double precision.  Every block cites the reference line it restates
 * (paths relative to /root/reference/).  [OpEn] marks statements about the third-party solver
 * (opengen 0.7.1 / Rust crate optimization_engine) that is absent from the reference tree:
 * those restate the published PANOC + ALM/PM algorithm and are the build's stated spec
 * ("parity unpinned" for the iteration).
 */
#define _POSIX_C_SOURCE 200809L
#include "mpc_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NMAX MPC_ORACLE_NMAX
#define NU_MAX (2 * NMAX)
#define NDYN_MAX 32
#define MEM_MAX 16

/* ------------------------------------------------------------------------------------------ */
/* parameter layout: src/mpc_traj_tracker/mpc/mpc_generator.py:179-188                          */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    int r0, c0, os0, od0, qs0, qd0, np;
} layout_t;

static layout_t make_layout(const mpc_oracle_config* cfg) {
    layout_t L;
    const int N = cfg->N;
    L.r0 = 8 + 10;                               /* s(8) + q(10) */
    L.c0 = L.r0 + 3 * N + N;                     /* r: N x (x,y,theta) then N speed refs */
    L.os0 = L.c0 + 3 * N * cfg->Nother;          /* c */
    L.od0 = L.os0 + cfg->Nstcobs * cfg->nstcobs; /* o_s */
    L.qs0 = L.od0 + cfg->Ndynobs * cfg->ndynobs * N; /* o_d */
    L.qd0 = L.qs0 + N;                           /* q_stc */
    L.np = L.qd0 + N;                            /* q_dyn */
    return L;
}

int32_t mpc_oracle_np(const mpc_oracle_config* cfg) { return make_layout(cfg).np; }

/* ------------------------------------------------------------------------------------------ */
/* unicycle RK4, literal form: src/pkg_motion_model/motion_model.py:151-164                     */
/* ------------------------------------------------------------------------------------------ */
static void d_state_f(const double s[3], const double a[2], double ts, double out[3]) {
    out[0] = ts * a[0] * cos(s[2]);
    out[1] = ts * a[0] * sin(s[2]);
    out[2] = ts * a[1];
}

void mpc_oracle_unicycle_rk4(const double s[3], const double a[2], double ts, double out[3]) {
    double k1[3], k2[3], k3[3], k4[3], t[3];
    d_state_f(s, a, ts, k1);
    for (int i = 0; i < 3; ++i) t[i] = s[i] + 0.5 * k1[i];
    d_state_f(t, a, ts, k2);
    for (int i = 0; i < 3; ++i) t[i] = s[i] + 0.5 * k2[i];
    d_state_f(t, a, ts, k3);
    for (int i = 0; i < 3; ++i) t[i] = s[i] + k3[i];
    d_state_f(t, a, ts, k4);
    for (int i = 0; i < 3; ++i) out[i] = s[i] + (1.0 / 6.0) * (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]);
}

/* ------------------------------------------------------------------------------------------ */
/* cost / constraints / gradient                                                               */
/* ------------------------------------------------------------------------------------------ */
static inline double sq(double x) { return x * x; }

/* squared distance of z to the box [lo,hi] and its derivative (opengen Rectangle.distance_squared) */
static inline double box_dist_sq(double z, double lo, double hi, double* dz) {
    if (z > hi) { *dz = 2.0 * (z - hi); return sq(z - hi); }
    if (z < lo) { *dz = 2.0 * (z - lo); return sq(z - lo); }
    *dz = 0.0;
    return 0.0;
}

void mpc_oracle_cost_grad(const mpc_oracle_config* cfg, const double* u, double c, const double* y,
                          const double* p, double* f_out, double* psi_out, double* grad, double* F1_out,
                          double* F2_out) {
    const int N = cfg->N;
    const layout_t L = make_layout(cfg);
    const double ts = cfg->ts;
    const int Nd = cfg->Ndynobs, Ns = cfg->Nstcobs, No = cfg->Nother;
    const int ne = cfg->nstcobs / 3; /* mpc_generator.py:221 */

    /* mpc_generator.py:190-192 */
    const double x_goal = p[3], y_goal = p[4], th_goal = p[5], v_init = p[6], w_init = p[7];
    const double qvel = p[9], rv = p[11], rw = p[12];
    const double qN = p[13], qthetaN = p[14], qrpd = p[15], acc_pen = p[16], w_acc_pen = p[17];

    /* ---- rollout (mpc_generator.py:199-204; RK4 of the unicycle collapses to Simpson's rule on
     *      the heading because theta' = w is state independent: SURVEY.md Appendix E) ------------ */
    double X[NMAX + 1], Y[NMAX + 1], TH[NMAX + 1];
    double Cx[NMAX], Sy[NMAX], dCw[NMAX], dSw[NMAX];
    X[0] = p[0]; Y[0] = p[1]; TH[0] = p[2];
    for (int k = 0; k < N; ++k) {
        const double v = u[2 * k], w = u[2 * k + 1];
        const double th0 = TH[k], thm = th0 + 0.5 * ts * w, th1 = th0 + ts * w;
        const double c0 = cos(th0), s0 = sin(th0), c1 = cos(thm), s1 = sin(thm), c2 = cos(th1), s2 = sin(th1);
        Cx[k] = (c0 + 4.0 * c1 + c2) / 6.0;
        Sy[k] = (s0 + 4.0 * s1 + s2) / 6.0;
        dCw[k] = -ts * (2.0 * s1 + s2) / 6.0; /* d Cx / d w at fixed theta_k */
        dSw[k] = ts * (2.0 * c1 + c2) / 6.0;
        X[k + 1] = X[k] + ts * v * Cx[k];
        Y[k + 1] = Y[k] + ts * v * Sy[k];
        TH[k + 1] = th1;
    }

    double f = 0.0;
    double Gx[NMAX], Gy[NMAX];             /* d f / d pos_{k+1}  (smooth cost part)        */
    double dSx[NMAX], dSy[NMAX];           /* d S / d pos_{k+1}  (static penalty)          */
    static const int DSTRIDE = NDYN_MAX;
    double dDx[NMAX * NDYN_MAX], dDy[NMAX * NDYN_MAX]; /* d D_i / d pos_{k+1}             */
    double S = 0.0, D[NDYN_MAX];
    for (int i = 0; i < Nd; ++i) D[i] = 0.0;
    double gu[NU_MAX];
    for (int i = 0; i < 2 * N; ++i) gu[i] = 0.0;

    for (int k = 0; k < N; ++k) {
        const double px = X[k + 1], py = Y[k + 1];
        const double v = u[2 * k], w = u[2 * k + 1];
        double gx = 0.0, gy = 0.0;

        /* -- reference path deviation: mpc_generator.py:207,116-130,28-36; path_ref[N]:=path_ref[N-1] (:194-195).
         *    dist**2 of a sqrt is the squared distance itself. min over segments i = k..N-1. */
        {
            double best = INFINITY, bgx = 0.0, bgy = 0.0;
            for (int i = k; i < N; ++i) {
                const int i2 = (i + 1 < N) ? i + 1 : N - 1;
                const double s1x = p[L.r0 + 3 * i], s1y = p[L.r0 + 3 * i + 1];
                const double s2x = p[L.r0 + 3 * i2], s2y = p[L.r0 + 3 * i2 + 1];
                const double dx = s2x - s1x, dy = s2y - s1y;
                const double den = dx * dx + dy * dy + 1e-16;
                const double th = ((px - s1x) * dx + (py - s1y) * dy) / den;
                const double t = fmin(fmax(th, 0.0), 1.0);
                const double wx = s1x + t * dx - px, wy = s1y + t * dy - py;
                const double d2 = wx * wx + wy * wy;
                if (d2 < best) { /* ties keep the earlier segment (fmin keeps its first operand) */
                    best = d2;
                    /* CasADi: d fmax(x,0)/dx = (x>=0), d fmin(x,1)/dx = (x<=1) */
                    const double pass = (th >= 0.0 && th <= 1.0) ? 1.0 : 0.0;
                    const double wd = (wx * dx + wy * dy) * pass / den;
                    bgx = 2.0 * (wd * dx - wx);
                    bgy = 2.0 * (wd * dy - wy);
                }
            }
            f += qrpd * best;
            gx += qrpd * bgx;
            gy += qrpd * bgy;
        }

        /* -- speed reference + control action: mpc_generator.py:208-209,97-99,110-111 */
        {
            const double vref = p[L.r0 + 3 * N + k];
            f += qvel * sq(v - vref) + rv * v * v + rw * w * w;
            gu[2 * k] += 2.0 * qvel * (v - vref) + 2.0 * rv * v;
            gu[2 * k + 1] += 2.0 * rw * w;
        }

        /* -- fleet collision: mpc_generator.py:211-216,105-108,25-26 */
        {
            const double W2 = cfg->vehicle_width * cfg->vehicle_width;
            for (int j = 0; j < No; ++j) {
                const double cx = p[L.c0 + j * 3 * N + 3 * k], cy = p[L.c0 + j * 3 * N + 3 * k + 1];
                const double ex = px - cx, ey = py - cy;
                const double h = W2 - (ex * ex + ey * ey);
                if (h > 0.0) {
                    f += cfg->fleet_weight * h;
                    gx += cfg->fleet_weight * (-2.0 * ex);
                    gy += cfg->fleet_weight * (-2.0 * ey);
                }
            }
        }

        /* -- static obstacles: mpc_generator.py:219-225,46-54 (hard part only; the soft cost is
         *    commented out at :227).  S accumulates over steps and obstacles. */
        {
            double sx = 0.0, sy = 0.0;
            for (int o = 0; o < Ns; ++o) {
                const double* b = p + L.os0 + o * cfg->nstcobs;
                const double* a0 = b + ne;
                const double* a1 = b + 2 * ne;
                double m[16];
                double prod = 1.0;
                for (int e = 0; e < ne; ++e) {
                    m[e] = fmax(0.0, b[e] - a0[e] * px - a1[e] * py);
                    prod *= m[e] * m[e];
                }
                if (prod > 0.0) { /* fmax(0, inside): derivative only where inside > 0 */
                    S += prod;
                    for (int e = 0; e < ne; ++e) {
                        const double rest = prod / (m[e] * m[e]); /* all m > 0 here */
                        sx += rest * 2.0 * m[e] * (-a0[e]);
                        sy += rest * 2.0 * m[e] * (-a1[e]);
                    }
                }
            }
            dSx[k] = sx;
            dSy[k] = sy;
        }

        /* -- dynamic obstacles: mpc_generator.py:229-241,38-44,85-95 */
        {
            const double qdyn = p[L.qd0 + k];
            for (int i = 0; i < Nd; ++i) {
                const double* e = p + L.od0 + i * cfg->ndynobs * N + cfg->ndynobs * k;
                const double cx = e[0], cy = e[1], rx = e[2], ry = e[3], ang = e[4], alpha = e[5];
                const double ca = cos(ang), sa = sin(ang);
                const double ex = px - cx, ey = py - cy;
                const double a = ex * ca + ey * sa, b = ex * sa - ey * ca;
                /* hard: F2_i += fmax(0, inside) (:238-239) */
                {
                    const double ix = 1.0 / sq(rx + 1e-6), iy = 1.0 / sq(ry + 1e-6);
                    const double I = 1.0 - a * a * ix - b * b * iy;
                    if (I > 0.0) {
                        D[i] += I;
                        dDx[k * DSTRIDE + i] = -2.0 * a * ca * ix - 2.0 * b * sa * iy;
                        dDy[k * DSTRIDE + i] = -2.0 * a * sa * ix + 2.0 * b * ca * iy;
                    } else {
                        dDx[k * DSTRIDE + i] = 0.0;
                        dDy[k * DSTRIDE + i] = 0.0;
                    }
                }
                /* soft: cost += q_dyn[k] * alpha * fmax(0, inside_with_margin)^2 (:241,:91-92) */
                {
                    const double ix = 1.0 / sq(rx + cfg->social_margin + 1e-6);
                    const double iy = 1.0 / sq(ry + cfg->social_margin + 1e-6);
                    const double I = 1.0 - a * a * ix - b * b * iy;
                    if (I > 0.0) {
                        f += qdyn * alpha * I * I;
                        const double wI = qdyn * alpha * 2.0 * I;
                        gx += wI * (-2.0 * a * ca * ix - 2.0 * b * sa * iy);
                        gy += wI * (-2.0 * a * sa * ix + 2.0 * b * ca * iy);
                    }
                }
            }
        }
        Gx[k] = gx;
        Gy[k] = gy;
    }

    /* -- terminal cost: mpc_generator.py:246 */
    double gthN = 0.0;
    {
        f += qN * (sq(X[N] - x_goal) + sq(Y[N] - y_goal)) + qthetaN * sq(TH[N] - th_goal);
        Gx[N - 1] += 2.0 * qN * (X[N] - x_goal);
        Gy[N - 1] += 2.0 * qN * (Y[N] - y_goal);
        gthN = 2.0 * qthetaN * (TH[N] - th_goal);
    }

    /* -- accelerations: mapping F1 and cost (mpc_generator.py:254-267) */
    double F1[NU_MAX];
    for (int k = 0; k < N; ++k) {
        const double vp = (k == 0) ? v_init : u[2 * (k - 1)];
        const double wp = (k == 0) ? w_init : u[2 * (k - 1) + 1];
        F1[k] = (u[2 * k] - vp) / ts;
        F1[N + k] = (u[2 * k + 1] - wp) / ts;
    }
    for (int k = 0; k < N; ++k) {
        f += acc_pen * sq(F1[k]) + w_acc_pen * sq(F1[N + k]);
        const double da = 2.0 * acc_pen * F1[k] / ts, db = 2.0 * w_acc_pen * F1[N + k] / ts;
        gu[2 * k] += da;
        gu[2 * k + 1] += db;
        if (k > 0) { gu[2 * (k - 1)] -= da; gu[2 * (k - 1) + 1] -= db; }
    }

    /* -- F2 = S (broadcast) + D_i : mpc_generator.py:198,225,239,272 */
    double F2[NDYN_MAX], sumF2 = 0.0, nrm2F2 = 0.0;
    for (int i = 0; i < Nd; ++i) {
        F2[i] = S + D[i];
        sumF2 += F2[i];
        nrm2F2 += F2[i] * F2[i];
    }

    /* -- psi [OpEn: opengen builder, psi = f + c/2 dist^2_C(F1 + y/max(c,1)) + c/2 ||F2||^2] */
    double psi = f;
    double dF1[NU_MAX]; /* d psi / d F1 */
    {
        const double cm = fmax(c, 1.0);
        double dist2 = 0.0;
        for (int j = 0; j < 2 * N; ++j) {
            const double lo = (j < N) ? cfg->lin_acc_min : -cfg->ang_acc_max; /* :260-264 */
            const double hi = (j < N) ? cfg->lin_acc_max : cfg->ang_acc_max;
            const double z = F1[j] + (y ? y[j] : 0.0) / cm;
            double dz;
            dist2 += box_dist_sq(z, lo, hi, &dz);
            dF1[j] = 0.5 * c * dz;
        }
        psi += 0.5 * c * dist2 + 0.5 * c * nrm2F2;
    }

    if (grad) {
        /* ALM term through F1 */
        for (int k = 0; k < N; ++k) {
            const double da = dF1[k] / ts, db = dF1[N + k] / ts;
            gu[2 * k] += da;
            gu[2 * k + 1] += db;
            if (k > 0) { gu[2 * (k - 1)] -= da; gu[2 * (k - 1) + 1] -= db; }
        }
        /* penalty term: c * sum_i F2_i * (dS + dD_i) */
        for (int k = 0; k < N; ++k) {
            double gx = Gx[k] + c * sumF2 * dSx[k], gy = Gy[k] + c * sumF2 * dSy[k];
            for (int i = 0; i < Nd; ++i) {
                gx += c * F2[i] * dDx[k * DSTRIDE + i];
                gy += c * F2[i] * dDy[k * DSTRIDE + i];
            }
            Gx[k] = gx;
            Gy[k] = gy;
        }
        /* adjoint sweep through the rollout (Jacobian of SURVEY.md Appendix E) */
        double lx = 0.0, ly = 0.0, lth = gthN;
        for (int k = N - 1; k >= 0; --k) {
            const double v = u[2 * k];
            lx += Gx[k];
            ly += Gy[k];
            gu[2 * k] += ts * (Cx[k] * lx + Sy[k] * ly);
            gu[2 * k + 1] += ts * v * (dCw[k] * lx + dSw[k] * ly) + ts * lth;
            lth += ts * v * (-Sy[k] * lx + Cx[k] * ly);
        }
        for (int i = 0; i < 2 * N; ++i) grad[i] = gu[i];
    }
    if (f_out) *f_out = f;
    if (psi_out) *psi_out = psi;
    if (F1_out) for (int j = 0; j < 2 * N; ++j) F1_out[j] = F1[j];
    if (F2_out) for (int i = 0; i < Nd; ++i) F2_out[i] = F2[i];
}

/* ------------------------------------------------------------------------------------------ */
/* [OpEn] L-BFGS buffer (crate `lbfgs`: newest pair at index 0, C-BFGS acceptance test)        */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    int n, mem, active, first_old;
    int gram; /* cfg->lbfgs_gram */
    double gamma;
    double s[MEM_MAX + 1][NU_MAX], y[MEM_MAX + 1][NU_MAX];
    double rho[MEM_MAX + 1], alpha[MEM_MAX + 1];
    double old_state[NU_MAX], old_g[NU_MAX];
    /* Gram form (lbfgs_gram != 0): sy[i][j] = s_i . y_j, yy[i][j] = y_i . y_j of the pairs held, newest at index 0 */
    double sy[MEM_MAX + 1][MEM_MAX + 1], yy[MEM_MAX + 1][MEM_MAX + 1];
    double dev_max; /* lbfgs_gram == 2: largest relative deviation of the Gram direction from the two-loop direction */
} lbfgs_t;

static const double CBFGS_EPSILON = 1e-8, SY_EPSILON = 1e-10; /* C-BFGS alpha = 1 */

static double dot(const double* a, const double* b, int n) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
}
static double norm2(const double* a, int n) { return sqrt(dot(a, a, n)); }

static void lbfgs_reset(lbfgs_t* l) { l->active = 0; l->first_old = 1; }

static void lbfgs_update(lbfgs_t* l, const double* g, const double* state) {
    const int n = l->n, m = l->mem;
    if (l->first_old) {
        l->first_old = 0;
        memcpy(l->old_state, state, n * sizeof(double));
        memcpy(l->old_g, g, n * sizeof(double));
        return;
    }
    double* sn = l->s[m];
    double* yn = l->y[m];
    for (int i = 0; i < n; ++i) { sn[i] = state[i] - l->old_state[i]; yn[i] = g[i] - l->old_g[i]; }
    const double ys = dot(sn, yn, n), ss = dot(sn, sn, n);
    if (ss <= DBL_MIN || ys <= SY_EPSILON) return;                          /* rejected */
    /* C-BFGS (Li-Fukushima): s'y / ||s||^2 > eps ||g||^alpha with alpha = 1; ||s||^2 > 0 was checked just above, so the
     * test is written as a multiplication -- the form the GPU kernel evaluates (mpc_kernels.hpp PanocLbfgs::update) */
    if (!(ys > (CBFGS_EPSILON * norm2(g, n)) * ss)) return;
    memcpy(l->old_state, state, n * sizeof(double));
    memcpy(l->old_g, g, n * sizeof(double));
    /* rotate right by one: the fresh pair moves to index 0 */
    double ts_[NU_MAX], ty_[NU_MAX];
    memcpy(ts_, sn, n * sizeof(double));
    memcpy(ty_, yn, n * sizeof(double));
    if (l->gram) {
        /* Gram matrices follow the rotation; the new row / column 0 holds the products with the pairs that stay */
        for (int i = m; i > 0; --i)
            for (int j = m; j > 0; --j) { l->sy[i][j] = l->sy[i - 1][j - 1]; l->yy[i][j] = l->yy[i - 1][j - 1]; }
        for (int j = 1; j <= m; ++j) {
            l->sy[0][j] = dot(ts_, l->y[j - 1], n);   /* s_new . y_j */
            l->sy[j][0] = dot(l->s[j - 1], ty_, n);   /* s_j . y_new */
            l->yy[0][j] = l->yy[j][0] = dot(ty_, l->y[j - 1], n);
        }
        l->sy[0][0] = ys;
        l->yy[0][0] = dot(ty_, ty_, n);
    }
    for (int j = m; j > 0; --j) {
        memcpy(l->s[j], l->s[j - 1], n * sizeof(double));
        memcpy(l->y[j], l->y[j - 1], n * sizeof(double));
        l->rho[j] = l->rho[j - 1];
    }
    memcpy(l->s[0], ts_, n * sizeof(double));
    memcpy(l->y[0], ty_, n * sizeof(double));
    l->rho[0] = 1.0 / ys;
    l->gamma = ys / dot(l->y[0], l->y[0], n);
    l->active = (l->active + 1 < m) ? l->active + 1 : m;
}

/* two-loop recursion, q <- H q [crate lbfgs: apply_hessian] */
static void lbfgs_apply_two_loop(lbfgs_t* l, double* q) {
    const int n = l->n;
    if (l->active == 0) return;
    for (int j = 0; j < l->active; ++j) {
        const double a = l->rho[j] * dot(l->s[j], q, n);
        l->alpha[j] = a;
        for (int i = 0; i < n; ++i) q[i] -= a * l->y[j][i];
    }
    for (int i = 0; i < n; ++i) q[i] *= l->gamma;
    for (int j = l->active - 1; j >= 0; --j) {
        const double b = l->rho[j] * dot(l->y[j], q, n);
        for (int i = 0; i < n; ++i) q[i] += (l->alpha[j] - b) * l->s[j][i];
    }
}

/* The same operator H in Gram form -- what the GPU kernel evaluates (mpc_kernels.hpp PanocLbfgs::direction): the 2m inner
 * products S'r, Y'r are independent of each other (one matrix-vector pass instead of 2m dependent reductions), the two loops
 * become scalar recurrences on the cached Gram entries s_i.y_j, y_i.y_j, and the direction is one linear combination
 *   d = gamma r + sum_p (alpha_p - beta_p) s_p - gamma alpha_p y_p.
 * Exact-arithmetic identical to the two-loop recursion: with q_p = r - sum_{q<p} alpha_q y_q,
 *   s_p.q_p = s_p.r - sum_{q<p} alpha_q (s_p.y_q)                                     (first loop, newest pair first)
 *   y_p.z_p = gamma (y_p.r - sum_q alpha_q (y_p.y_q)) + sum_{q>p} delta_q (s_q.y_p)   (second loop, oldest pair first) */
static void lbfgs_apply_gram(lbfgs_t* l, double* q) {
    const int n = l->n, a = l->active;
    if (a == 0) return;
    double acc[MEM_MAX + 1], t[MEM_MAX + 1], alpha[MEM_MAX + 1], delta[MEM_MAX + 1];
    for (int j = 0; j < a; ++j) { acc[j] = dot(l->s[j], q, n); t[j] = dot(l->y[j], q, n); }
    for (int p = 0; p < a; ++p) {
        alpha[p] = l->rho[p] * acc[p];
        for (int j = 0; j < a; ++j) { acc[j] -= alpha[p] * l->sy[j][p]; t[j] -= alpha[p] * l->yy[j][p]; }
    }
    for (int j = 0; j < a; ++j) t[j] *= l->gamma;
    for (int p = a - 1; p >= 0; --p) {
        delta[p] = alpha[p] - l->rho[p] * t[p];
        for (int j = 0; j < a; ++j) t[j] += delta[p] * l->sy[p][j];
    }
    for (int i = 0; i < n; ++i) {
        double d = l->gamma * q[i];
        for (int p = 0; p < a; ++p) d += delta[p] * l->s[p][i];
        for (int p = 0; p < a; ++p) d += (-(l->gamma * alpha[p])) * l->y[p][i];
        q[i] = d;
    }
}

static void lbfgs_apply(lbfgs_t* l, double* q) {
    if (l->gram == 1) { lbfgs_apply_gram(l, q); return; }
    if (l->gram == 2 && l->active > 0) { /* two-loop drives the iteration; the Gram form is evaluated beside it */
        double g[NU_MAX];
        memcpy(g, q, l->n * sizeof(double));
        lbfgs_apply_gram(l, g);
        lbfgs_apply_two_loop(l, q);
        double num = 0.0, den = 0.0;
        for (int i = 0; i < l->n; ++i) { num = fmax(num, fabs(g[i] - q[i])); den = fmax(den, fabs(q[i])); }
        if (den > 0.0 && num / den > l->dev_max) l->dev_max = num / den;
        return;
    }
    lbfgs_apply_two_loop(l, q);
}

/* ------------------------------------------------------------------------------------------ */
/* [OpEn] PANOC inner solver                                                                   */
/* ------------------------------------------------------------------------------------------ */
static const double GAMMA_L_COEFF = 0.95;
static const double DELTA_LIPSCHITZ = 1e-12, EPSILON_LIPSCHITZ = 1e-6;
static const double LIPSCHITZ_UPDATE_EPSILON = 1e-6;
static const int MAX_LIPSCHITZ_UPDATE_ITERATIONS = 10;
static const double MAX_LIPSCHITZ_CONSTANT = 1e9, MIN_L_ESTIMATE = 1e-10;
static const int MAX_LINESEARCH_ITERATIONS = 10;

typedef struct {
    const mpc_oracle_config* cfg;
    const double* p;
    double c;
    const double* y;
    int n;
    int n_cost, n_grad;
    /* cache */
    lbfgs_t lb;
    double grad[NU_MAX], grad_prev[NU_MAX], u_half[NU_MAX], gstep[NU_MAX], dir[NU_MAX], u_plus[NU_MAX],
        gfpr[NU_MAX];
    double gamma, L, sigma, cost, norm_gfpr, tau, rhs_ls, lhs_ls;
    int iter;
    double tol, akkt_tol;
    struct timespec t0;
    double max_us;
    /* decision trace (mpc_oracle_solve_trace) */
    double* trace;
    int trace_cap, trace_n, outer;
} panoc_t;

static double elapsed_us(const struct timespec* t0) {
    struct timespec t1;
    clock_gettime(CLOCK_MONOTONIC, &t1);
    return (t1.tv_sec - t0->tv_sec) * 1e6 + (t1.tv_nsec - t0->tv_nsec) * 1e-3;
}

static double eval_cost(panoc_t* s, const double* u) {
    double psi;
    mpc_oracle_cost_grad(s->cfg, u, s->c, s->y, s->p, NULL, &psi, NULL, NULL, NULL);
    s->n_cost++;
    return psi;
}
static double eval_cost_grad(panoc_t* s, const double* u, double* g) {
    double psi;
    mpc_oracle_cost_grad(s->cfg, u, s->c, s->y, s->p, NULL, &psi, g, NULL, NULL);
    s->n_cost++;
    s->n_grad++;
    return psi;
}

/* Rectangle::project on U: mpc_generator.py:249-251 */
static void project_U(const mpc_oracle_config* cfg, double* u, int N) {
    for (int k = 0; k < N; ++k) {
        u[2 * k] = fmin(fmax(u[2 * k], cfg->lin_vel_min), cfg->lin_vel_max);
        u[2 * k + 1] = fmin(fmax(u[2 * k + 1], -cfg->ang_vel_max), cfg->ang_vel_max);
    }
}

static void gradient_step(panoc_t* s, const double* u) {
    for (int i = 0; i < s->n; ++i) s->gstep[i] = u[i] - s->gamma * s->grad[i];
}
static void half_step(panoc_t* s) {
    memcpy(s->u_half, s->gstep, s->n * sizeof(double));
    project_U(s->cfg, s->u_half, s->cfg->N);
}
static void compute_fpr(panoc_t* s, const double* u) {
    for (int i = 0; i < s->n; ++i) s->gfpr[i] = u[i] - s->u_half[i];
    s->norm_gfpr = norm2(s->gfpr, s->n);
}
static void panoc_cache_reset(panoc_t* s) {
    lbfgs_reset(&s->lb);
    s->lhs_ls = s->rhs_ls = 0.0;
    s->tau = 1.0;
    s->L = s->sigma = s->cost = s->gamma = 0.0;
    s->iter = 0;
}
static void set_akkt_tolerance(panoc_t* s, double t) {
    s->akkt_tol = t;
    memset(s->grad_prev, 0, sizeof(s->grad_prev));
}

static void panoc_init(panoc_t* s, double* u) {
    const int n = s->n;
    panoc_cache_reset(s);
    /* cost + gradient at u, local Lipschitz estimate of grad psi by one finite difference:
     * h_i = max(delta, eps*u_i), L = ||grad(u+h) - grad(u)|| / ||h|| */
    s->cost = eval_cost_grad(s, u, s->grad);
    double h[NU_MAX], uh[NU_MAX], gh[NU_MAX];
    for (int i = 0; i < n; ++i) {
        h[i] = (EPSILON_LIPSCHITZ * u[i] > DELTA_LIPSCHITZ) ? EPSILON_LIPSCHITZ * u[i] : DELTA_LIPSCHITZ;
        uh[i] = u[i] + h[i];
    }
    eval_cost_grad(s, uh, gh);
    s->n_cost--; /* only the gradient is needed here */
    for (int i = 0; i < n; ++i) gh[i] -= s->grad[i];
    s->L = norm2(gh, n) / norm2(h, n);
    s->gamma = GAMMA_L_COEFF / fmax(s->L, MIN_L_ESTIMATE);
    s->sigma = (1.0 - GAMMA_L_COEFF) / (4.0 * s->gamma);
    gradient_step(s, u);
    half_step(s);
}

static int exit_condition(panoc_t* s) {
    if (!(s->norm_gfpr < s->tol)) return 0;
    /* AKKT residual || gfpr/gamma + grad - grad_prev || < eps_nu */
    double r = 0.0;
    for (int i = 0; i < s->n; ++i) r += sq(s->gfpr[i] / s->gamma + s->grad[i] - s->grad_prev[i]);
    return sqrt(r) < s->akkt_tol;
}

static int update_lipschitz(panoc_t* s, const double* u) {
    double cost_half = eval_cost(s, s->u_half);
    /* s->cost already holds psi(u) (the crate re-evaluates it; same value) */
    int it = 0;
    for (;;) {
        const double ip = dot(s->grad, s->gfpr, s->n);
        const double rhs = s->cost + LIPSCHITZ_UPDATE_EPSILON * fabs(s->cost) - ip +
                           (GAMMA_L_COEFF / (2.0 * s->gamma)) * sq(s->norm_gfpr);
        if (!(cost_half > rhs && it < MAX_LIPSCHITZ_UPDATE_ITERATIONS && s->L < MAX_LIPSCHITZ_CONSTANT)) break;
        lbfgs_reset(&s->lb);
        s->L *= 2.0;
        s->gamma /= 2.0;
        gradient_step(s, u);
        half_step(s);
        cost_half = eval_cost(s, s->u_half);
        compute_fpr(s, u);
        ++it;
    }
    s->sigma = (1.0 - GAMMA_L_COEFF) / (4.0 * s->gamma);
    return it;
}

static double fbe(const panoc_t* s) {
    /* forward-backward envelope from the cached cost / grad / gstep / u_half */
    double gg = 0.0, d2 = 0.0;
    for (int i = 0; i < s->n; ++i) { gg += sq(s->grad[i]); d2 += sq(s->gstep[i] - s->u_half[i]); }
    return s->cost - 0.5 * s->gamma * gg + 0.5 * d2 / s->gamma;
}

/* returns 1 to keep iterating */
static int panoc_step(panoc_t* s, double* u) {
    const int n = s->n;
    if (s->iter >= 1) memcpy(s->grad_prev, s->grad, n * sizeof(double));
    compute_fpr(s, u);
    if (exit_condition(s)) return 0;
    const int n_lip = update_lipschitz(s, u);
    const double psi_u = s->cost, nfpr_u = s->norm_gfpr;
    int nls = -1;
    /* L-BFGS: buffer update with (state = u, g = gamma*fpr), direction = H * gfpr */
    lbfgs_update(&s->lb, s->gfpr, u);
    if (s->iter > 0) {
        memcpy(s->dir, s->gfpr, n * sizeof(double));
        lbfgs_apply(&s->lb, s->dir);
    }
    if (s->iter == 0) {
        memcpy(u, s->u_half, n * sizeof(double));
        s->cost = eval_cost_grad(s, u, s->grad);
        gradient_step(s, u);
        half_step(s);
    } else {
        s->rhs_ls = fbe(s) - s->sigma * sq(s->norm_gfpr);
        s->tau = 1.0;
        nls = 0;
        for (;;) {
            for (int i = 0; i < n; ++i) s->u_plus[i] = u[i] - (1.0 - s->tau) * s->gfpr[i] - s->tau * s->dir[i];
            s->cost = eval_cost_grad(s, s->u_plus, s->grad);
            for (int i = 0; i < n; ++i) s->gstep[i] = s->u_plus[i] - s->gamma * s->grad[i];
            half_step(s);
            s->lhs_ls = fbe(s);
            if (!(s->lhs_ls > s->rhs_ls && nls < MAX_LINESEARCH_ITERATIONS)) break;
            s->tau /= 2.0;
            ++nls;
        }
        /* MAX_LINESEARCH_ITERATIONS halvings without acceptance.  ls_fallback = 0: the last trial point (tau = 2^-10) is
         * the next iterate -- in the published code the `tau = 0; u <- u_half` fallback is immediately overwritten by the
         * copy of u_plus into u, so this is what it effectively does.  ls_fallback = 1: the fallback as SURVEY.md
         * Appendix B words it: tau = 0, the point u - gamma*fpr (the half step of u) is evaluated and taken. */
        if (s->cfg->ls_fallback == 1 && s->lhs_ls > s->rhs_ls) {
            s->tau = 0.0;
            for (int i = 0; i < n; ++i) s->u_plus[i] = u[i] - s->gfpr[i];
            s->cost = eval_cost_grad(s, s->u_plus, s->grad);
            for (int i = 0; i < n; ++i) s->gstep[i] = s->u_plus[i] - s->gamma * s->grad[i];
            half_step(s);
        }
        memcpy(u, s->u_plus, n * sizeof(double));
    }
    if (s->trace) {
        if (s->trace_n < s->trace_cap) {
            double* r = s->trace + (size_t)s->trace_n * MPC_ORACLE_TRACE_FIELDS;
            r[0] = s->outer; r[1] = s->iter; r[2] = s->c; r[3] = s->L; r[4] = s->gamma; r[5] = nfpr_u; r[6] = psi_u;
            r[7] = n_lip; r[8] = s->lb.active; r[9] = nls; r[10] = s->tau; r[11] = s->cost;
        }
        s->trace_n++;
    }
    s->iter++;
    return 1;
}

/* returns exit status; *iters = loop count */
static int panoc_solve(panoc_t* s, double* u, int max_iter, int* iters) {
    panoc_init(s, u);
    int num_iter = 0, cont_iters = 1, cont_time = 1;
    int flag = panoc_step(s, u);
    while (flag && cont_iters && cont_time) {
        num_iter++;
        cont_iters = num_iter < max_iter;
        if (s->max_us > 0.0) cont_time = elapsed_us(&s->t0) <= s->max_us;
        flag = panoc_step(s, u);
    }
    memcpy(u, s->u_half, s->n * sizeof(double)); /* the feasible half step is returned */
    *iters = num_iter;
    if (!cont_iters) return MPC_ORACLE_NOTCONV_ITERS;
    if (!cont_time) return MPC_ORACLE_NOTCONV_TIME;
    return MPC_ORACLE_CONVERGED;
}

/* ------------------------------------------------------------------------------------------ */
/* [OpEn] ALM / penalty-method outer loop                                                      */
/* ------------------------------------------------------------------------------------------ */
static int32_t solve_impl(const mpc_oracle_config* cfg, const double* p, const double* u0, const double* y0,
                          double c0, double* u_out, double* y_out, mpc_oracle_result* res, double* trace,
                          int32_t cap, int32_t* n_steps) {
    const int N = cfg->N, n = 2 * N, n1 = 2 * N, n2 = cfg->Ndynobs;
    if (N > NMAX || N < 1 || cfg->lbfgs_mem > MEM_MAX || cfg->lbfgs_mem < 1 || n2 > NDYN_MAX) return -1;
    const double SMALL_EPSILON = DBL_EPSILON;
    panoc_t* s = (panoc_t*)calloc(1, sizeof(panoc_t));
    if (!s) return -2;
    s->cfg = cfg; s->p = p; s->n = n;
    s->lb.n = n; s->lb.mem = cfg->lbfgs_mem; s->lb.gram = cfg->lbfgs_gram; s->lb.dev_max = 0.0;
    s->tol = cfg->tol;
    s->max_us = cfg->max_duration_us;
    s->trace = trace; s->trace_cap = cap; s->trace_n = 0;
    clock_gettime(CLOCK_MONOTONIC, &s->t0);

    double u[NU_MAX], y[NU_MAX], y_plus[NU_MAX], F1[NU_MAX], F2[NDYN_MAX];
    for (int i = 0; i < n; ++i) u[i] = u0 ? u0[i] : 0.0;
    for (int i = 0; i < n1; ++i) { y[i] = y0 ? y0[i] : 0.0; y_plus[i] = y[i]; }
    double c = (c0 > 0.0) ? c0 : cfg->init_penalty;
    s->y = y;

    panoc_cache_reset(s);
    set_akkt_tolerance(s, cfg->init_tol);
    int iteration = 0, num_outer = 0, inner_total = 0, status = MPC_ORACLE_CONVERGED;
    double f2_norm = 0.0, f2_norm_plus = 0.0, dy_norm = 0.0, dy_norm_plus = 0.0, last_fpr = 0.0;

    for (int outer = 0; outer < cfg->max_outer; ++outer) {
        if (s->max_us > 0.0 && elapsed_us(&s->t0) > s->max_us) { status = MPC_ORACLE_NOTCONV_TIME; break; }
        num_outer++;
        /* y <- Proj_Y(y), Y = [-1e12, 1e12]^n1 */
        for (int i = 0; i < n1; ++i) y[i] = fmin(fmax(y[i], -1e12), 1e12);
        s->c = c;
        s->outer = outer;
        int it = 0;
        const int inner_status = panoc_solve(s, u, cfg->max_inner, &it);
        inner_total += it;
        last_fpr = s->norm_gfpr;
        /* y+ <- y + c (F1(u) - Proj_C(F1(u) + y/c)) */
        mpc_oracle_cost_grad(cfg, u, c, y, p, NULL, NULL, NULL, F1, F2);
        double dy2 = 0.0, f22 = 0.0;
        for (int j = 0; j < n1; ++j) {
            const double lo = (j < N) ? cfg->lin_acc_min : -cfg->ang_acc_max;
            const double hi = (j < N) ? cfg->lin_acc_max : cfg->ang_acc_max;
            const double z = F1[j] + y[j] / c;
            const double pz = fmin(fmax(z, lo), hi);
            y_plus[j] = y[j] + c * (F1[j] - pz);
            dy2 += sq(y_plus[j] - y[j]);
        }
        for (int i = 0; i < n2; ++i) f22 += sq(F2[i]);
        dy_norm_plus = sqrt(dy2);
        f2_norm_plus = sqrt(f22);
        /* exit criterion */
        const int crit1 = (n1 == 0) || (iteration > 0 && dy_norm_plus <= c * cfg->delta_tol + SMALL_EPSILON);
        const int crit2 = (n2 == 0) || (f2_norm_plus <= cfg->delta_tol + SMALL_EPSILON);
        const int crit3 = s->akkt_tol <= cfg->tol + SMALL_EPSILON;
        if (crit1 && crit2 && crit3) { status = inner_status; break; }
        /* penalty update unless the stall criterion holds: first iteration, or sufficient decrease of EITHER infeasibility
         * (stall_rule = 0: the published crate as recalled) / of BOTH (stall_rule = 1: SURVEY.md Appendix B) */
        const int alm_shrank = (n1 > 0) && dy_norm_plus <= cfg->suff_decrease * dy_norm + SMALL_EPSILON;
        const int pm_shrank = (n2 > 0) && f2_norm_plus <= cfg->suff_decrease * f2_norm + SMALL_EPSILON;
        const int stall = (iteration == 0) ||
                          (cfg->stall_rule == 1 ? ((n1 == 0 || alm_shrank) && (n2 == 0 || pm_shrank))
                                                : (alm_shrank || pm_shrank));
        if (!stall) c *= cfg->penalty_update;
        set_akkt_tolerance(s, fmax(s->akkt_tol * cfg->tol_update, cfg->tol));
        iteration++;
        dy_norm = dy_norm_plus;
        f2_norm = f2_norm_plus;
        memcpy(y, y_plus, n1 * sizeof(double));
        panoc_cache_reset(s);
    }
    if (status != MPC_ORACLE_NOTCONV_TIME && num_outer == cfg->max_outer) status = MPC_ORACLE_NOTCONV_ITERS;

    double cost = 0.0;
    mpc_oracle_cost_grad(cfg, u, 0.0, y, p, NULL, &cost, NULL, NULL, NULL); /* psi with c = 0 is f */
    for (int i = 0; i < n; ++i) u_out[i] = u[i];
    if (y_out) for (int i = 0; i < n1; ++i) y_out[i] = y_plus[i];
    if (res) {
        res->cost = cost;
        res->fpr = last_fpr;
        res->f2_norm = f2_norm_plus;
        res->delta_y_norm = dy_norm_plus / c;
        res->penalty = c;
        res->solve_time_ms = elapsed_us(&s->t0) * 1e-3;
        res->status = status;
        res->outer_iters = num_outer;
        res->inner_iters = inner_total;
        res->n_cost_evals = s->n_cost;
        res->n_grad_evals = s->n_grad;
        res->_pad = 0;
        res->lbfgs_dev = s->lb.dev_max;
    }
    if (n_steps) *n_steps = s->trace_n;
    free(s);
    return 0;
}

int32_t mpc_oracle_solve(const mpc_oracle_config* cfg, const double* p, const double* u0, const double* y0,
                         double c0, double* u_out, double* y_out, mpc_oracle_result* res) {
    return solve_impl(cfg, p, u0, y0, c0, u_out, y_out, res, NULL, 0, NULL);
}

int32_t mpc_oracle_solve_trace(const mpc_oracle_config* cfg, const double* p, const double* u0, const double* y0,
                               double c0, double* u_out, double* y_out, mpc_oracle_result* res, double* trace,
                               int32_t cap, int32_t* n_steps) {
    return solve_impl(cfg, p, u0, y0, c0, u_out, y_out, res, trace, cap, n_steps);
}

int32_t mpc_oracle_solve_batch(const mpc_oracle_config* cfg, int32_t B, const double* p, const double* u0,
                               const double* y0, const double* c0, double* u_out, double* y_out,
                               mpc_oracle_result* res, int32_t nthreads) {
    const int np = mpc_oracle_np(cfg), n = 2 * cfg->N;
    int used = 1;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
    used = nthreads;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
#endif
    for (int b = 0; b < B; ++b) {
        mpc_oracle_solve(cfg, p + (size_t)b * np, u0 ? u0 + (size_t)b * n : NULL, y0 ? y0 + (size_t)b * n : NULL,
                         c0 ? c0[b] : 0.0, u_out + (size_t)b * n, y_out ? y_out + (size_t)b * n : NULL,
                         res ? res + b : NULL);
    }
    return used;
}
