/*
 * Plain-C host of libmpcgpu.so (no Python, no torch): the C-ABI of include/mpcgpu.h used the way a C / C++ / cgo / JNI
 * binding would.  Builds B copies of a straight-corridor tracking problem in the reference's parameter layout
 * (src/mpc_traj_tracker/mpc/mpc_generator.py:179-188), solves them, prints the first input of each.
 *
 *   gcc -O2 -I include examples/solve_batch.c -o solve_batch -L trajtrack_mpcndqn_rlboost_amd -lmpcgpu \
 *       -Wl,-rpath,$PWD/trajtrack_mpcndqn_rlboost_amd
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "mpcgpu.h"

static double now_ms(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return 1e3 * (double)t.tv_sec + 1e-6 * (double)t.tv_nsec;
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 4;
    mpcgpu_config c;
    memset(&c, 0, sizeof c);
    c.N = 20; c.nu = 2; c.ns = 3; c.Nother = 10; c.Nstcobs = 10; c.nstcobs = 12; c.Ndynobs = 15; c.ndynobs = 6;
    c.ts = 0.2;
    c.lin_vel_min = -0.5; c.lin_vel_max = 1.5; c.ang_vel_max = 0.5;
    c.lin_acc_min = -1.0; c.lin_acc_max = 1.0; c.ang_acc_max = 3.0;
    c.vehicle_width = 0.5; c.social_margin = 0.2; c.fleet_weight = 1000.0;
    c.tol = 1e-4; c.delta_tol = 1e-4; c.init_tol = 1e-4; c.init_penalty = 10.0; c.penalty_update = 5.0;
    c.tol_update = 0.1; c.suff_decrease = 0.1;
    c.max_inner = 500; c.max_outer = 10; c.lbfgs_mem = 10; c.device = 0; c.max_duration_us = 5e6;

    void* h = NULL;
    if (mpcgpu_create(&c, &h) != 0) { fprintf(stderr, "create: %s\n", mpcgpu_last_error(NULL)); return 2; }
    const int np = mpcgpu_num_params(h), n = 2 * c.N;
    double* p = calloc((size_t)B * np, sizeof(double));
    double* u = malloc((size_t)B * n * sizeof(double));
    double* cost = malloc((size_t)B * sizeof(double));
    int32_t* status = malloc((size_t)B * sizeof(int32_t));
    int32_t* inner = malloc((size_t)B * sizeof(int32_t));
    for (int b = 0; b < B; ++b) {
        double* q = p + (size_t)b * np;
        const double y = 3.5 + 0.1 * b;
        q[0] = 0.6; q[1] = y; q[2] = 0.0;                       /* state */
        q[3] = 0.6 + 0.24 * c.N; q[4] = y; q[5] = 0.0;          /* horizon goal */
        q[6] = 1.0; q[7] = 0.0;                                 /* last input */
        const double w[10] = {0, 10, 0, 0, 0, 0, 0, 100, 10, 20};  /* qpos qvel qtheta rv rw qN qthetaN qrpd acc w_acc */
        memcpy(q + 8, w, sizeof w);
        for (int k = 0; k < c.N; ++k) {                         /* reference samples, then speed references */
            q[18 + 3 * k] = 0.6 + 0.24 * (k + 1); q[18 + 3 * k + 1] = y; q[18 + 3 * k + 2] = 0.0;
            q[18 + 3 * c.N + k] = 1.2;
        }
        for (int k = 0; k < 2 * c.N; ++k) q[np - 2 * c.N + k] = 1000.0;   /* q_stc, q_dyn */
    }
    const int rc = mpcgpu_solve_batch(h, B, p, NULL, NULL, NULL, u, cost, status, inner, NULL, NULL, NULL, NULL, NULL);
    if (rc != 0) { fprintf(stderr, "solve: %s\n", mpcgpu_last_error(h)); return 3; }
    int bad = 0;
    for (int b = 0; b < B; ++b) {
        printf("problem %d: status %d, %d inner iterations, cost %.6f, u0 = (%.6f, %.6f)\n", b, status[b], inner[b],
               cost[b], u[(size_t)b * n], u[(size_t)b * n + 1]);
        /* a free straight corridor at the reference speed: converged, accelerating towards 1.2 m/s, no turning */
        if (status[b] != MPCGPU_CONVERGED || !(u[(size_t)b * n] > 1.0 && u[(size_t)b * n] <= 1.2 + 1e-9) ||
            !(u[(size_t)b * n + 1] > -1e-6 && u[(size_t)b * n + 1] < 1e-6)) bad = 1;
    }
    /* what one tick of a single-robot user costs: the same call again, timed from host buffers in to results out */
    double best = 1e30;
    for (int r = 0; r < 10; ++r) {
        const double t0 = now_ms();
        if (mpcgpu_solve_batch(h, B, p, NULL, NULL, NULL, u, cost, status, inner, NULL, NULL, NULL, NULL, NULL) != 0) return 3;
        const double dt = now_ms() - t0;
        if (dt < best) best = dt;
    }
    double prep_ms = 0, solve_ms = 0;
    mpcgpu_last_timing(h, &prep_ms, &solve_ms);
    printf("kernel time: prep %.3f ms, solve %.3f ms; %d wavefronts per SIMD; latency kernel %d\n", prep_ms, solve_ms,
           mpcgpu_last_waves_per_simd(h), mpcgpu_last_latency_kernel(h));
    printf("call (host buffers in, results out), best of 10: %.3f ms for %d problem(s)\n", best, B);
    mpcgpu_destroy(h);
    free(p); free(u); free(cost); free(status); free(inner);
    return bad;
}
