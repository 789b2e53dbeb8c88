/*
 * mpc_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * Plain-C, double-precision restatement of the reference's NMPC hot path:
 *   - problem definition  : /root/reference/src/mpc_traj_tracker/mpc/mpc_generator.py:25-54,85-130,160-272
 *   - unicycle RK4 step   : /root/reference/src/pkg_motion_model/motion_model.py:142-164
 *   - solver algorithm    : NOT in /root/reference.  The reference generates it with the
 *                           third-party packages opengen==0.7.1 (requirements.txt:26) +
 *                           Rust crate `optimization_engine` (version unpinned: no Cargo.lock
 *                           in the reference) + casadi==3.5.5 (requirements.txt:1).  The
 *                           published algorithm (PANOC: Stella et al., CDC 2017; ALM/PM outer
 *                           loop: Sopasakis et al., IFAC 2020) is restated here; constants the
 *                           reference sets itself are taken from mpc_generator.py:22,285-293.
 *
 * PARITY STATUS: the cost / gradient / constraint mappings are pinned against fixtures produced by
 * executing the reference's own mpc_generator.py (tests/golden/make_fixtures.py).  The solver
 * ITERATION (PANOC/ALM) is "parity unpinned": the reference holds no golden vectors for it and the
 * real OpEn binary cannot be built in this image (no cargo/rustc/casadi/opengen).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#ifndef MPC_ORACLE_H
#define MPC_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPC_ORACLE_NMAX 64 /* max horizon handled by the oracle */

typedef struct mpc_oracle_config {
    /* problem dimensions: config/mpc_default.yaml:25,42-49 of the reference */
    int32_t N;        /* N_hor   */
    int32_t Nother;   /* max other robots */
    int32_t Nstcobs;  /* max static obstacles */
    int32_t nstcobs;  /* params per static obstacle (3 * n_edges) */
    int32_t Ndynobs;  /* max dynamic obstacles */
    int32_t ndynobs;  /* params per dynamic obstacle per step (6) */
    double ts;
    double lin_vel_min, lin_vel_max, ang_vel_max;
    double lin_acc_min, lin_acc_max, ang_acc_max;
    double vehicle_width, social_margin, fleet_weight; /* fleet_weight = 1000, mpc_generator.py:216 */
    /* solver hyper-parameters: mpc_generator.py:285-293 (+ opengen defaults) */
    double tol;            /* epsilon            1e-4 */
    double delta_tol;      /* delta              1e-4 */
    double init_tol;       /* epsilon_0          1e-4 */
    double init_penalty;   /* c_0                10   */
    double penalty_update; /* rho                5    */
    double tol_update;     /* beta               0.1  */
    double suff_decrease;  /* theta              0.1  */
    int32_t max_inner;     /* 500 */
    int32_t max_outer;     /* 10  */
    int32_t lbfgs_mem;     /* 10  */
    int32_t ls_fallback;   /* line search that fails 10 halvings: 0 = the last trial point (tau = 2^-10) is the next iterate
                              [what the published code does: its tau = 0 copy is overwritten by the u_plus swap];
                              1 = tau = 0, i.e. u - gamma*fpr is evaluated and taken */
    int32_t lbfgs_gram;    /* how H * (gamma fpr) is evaluated: 0 = two-loop recursion [the published crate]; 1 = the Gram form of
                              the same operator (what the GPU kernel evaluates, mpc_kernels.hpp PanocLbfgs::direction: identical in
                              exact arithmetic, rounded differently); 2 = two-loop drives the iteration, the Gram form is evaluated
                              beside it and the largest relative deviation is reported in mpc_oracle_result.lbfgs_dev */
    int32_t stall_rule;    /* when the penalty is kept ("penalty stall criterion"): 0 = in the first outer iteration or when EITHER
                              infeasibility shrank by theta [the published crate's is_penalty_stall_criterion as recalled:
                              iteration == 0 || (n1 > 0 && dy+ <= theta dy + eps) || (n2 > 0 && ||F2+|| <= theta ||F2|| + eps)];
                              1 = only when BOTH shrank (the reading of rounds 1-5) */
    double max_duration_us; /* 5e6; <=0 disables the wall-clock test */
} mpc_oracle_config;

/* exit_status codes (names: config/mpc_default.yaml:54 of the reference) */
enum { MPC_ORACLE_CONVERGED = 0, MPC_ORACLE_NOTCONV_ITERS = 1, MPC_ORACLE_NOTCONV_TIME = 2 };

/* number of parameters of one problem (len(p)), mpc_generator.py:179-188 */
int32_t mpc_oracle_np(const mpc_oracle_config* cfg);

/* One RK4 unicycle step, literal 4-stage form (motion_model.py:142-164). */
void mpc_oracle_unicycle_rk4(const double s[3], const double a[2], double ts, double out[3]);

/*
 * f, psi, grad psi, F1, F2 at (u; xi=(c,y); p).  Any output pointer may be NULL.
 *   psi = f + c/2 * dist^2_C(F1 + y/max(c,1)) + c/2 * ||F2||^2        [opengen psi construction]
 *   y may be NULL (treated as 0).
 */
void mpc_oracle_cost_grad(const mpc_oracle_config* cfg, const double* u, double c, const double* y,
                          const double* p, double* f, double* psi, double* grad, double* F1, double* F2);

typedef struct mpc_oracle_result {
    double cost;          /* f(u*) (psi evaluated with c = 0) */
    double fpr;           /* last inner ||gamma*fpr|| */
    double f2_norm;       /* ||F2(u*)|| */
    double delta_y_norm;  /* ||y+ - y|| / c */
    double penalty;       /* final c */
    double solve_time_ms;
    int32_t status;
    int32_t outer_iters;
    int32_t inner_iters;
    int32_t n_cost_evals; /* psi evaluations actually executed   */
    int32_t n_grad_evals; /* grad psi evaluations actually executed */
    int32_t _pad;
    double lbfgs_dev;     /* lbfgs_gram == 2: max over the solve of |d_gram - d_two_loop|_inf / |d_two_loop|_inf */
} mpc_oracle_result;

/* One solve. u0 / y0 may be NULL (zeros); c0 <= 0 means cfg->init_penalty. y_out may be NULL. */
int32_t mpc_oracle_solve(const mpc_oracle_config* cfg, const double* p, const double* u0, const double* y0,
                         double c0, double* u_out, double* y_out, mpc_oracle_result* res);

/*
 * Decision trace of one solve (test infrastructure for the GPU kernel's trace build): one record of
 * MPC_ORACLE_TRACE_FIELDS doubles per completed PANOC step, across all inner problems, first `cap` steps:
 *   0 outer index   1 step index inside the inner problem   2 penalty c   3 Lipschitz estimate L (after back-tracking)
 *   4 gamma         5 ||gamma*fpr|| at u                    6 psi(u)      7 Lipschitz doublings in this step
 *   8 L-BFGS pairs held after the buffer update             9 line-search halvings (-1: first step, no line search)
 *   10 accepted tau (1 on the first step; 0 after the fallback of ls_fallback = 1)          11 psi(u_next)
 * Returns the number of PANOC steps taken (may exceed cap) in *n_steps.
 */
#define MPC_ORACLE_TRACE_FIELDS 12
int32_t mpc_oracle_solve_trace(const mpc_oracle_config* cfg, const double* p, const double* u0, const double* y0,
                               double c0, double* u_out, double* y_out, mpc_oracle_result* res, double* trace,
                               int32_t cap, int32_t* n_steps);

/* B independent solves, OpenMP over the batch with `nthreads` threads (<=0: all).  Returns threads used. */
int32_t mpc_oracle_solve_batch(const mpc_oracle_config* cfg, int32_t B, const double* p, const double* u0,
                               const double* y0, const double* c0, double* u_out, double* y_out,
                               mpc_oracle_result* res, int32_t nthreads);

#ifdef __cplusplus
}
#endif
#endif
