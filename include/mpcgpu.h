/*
 * mpcgpu.h -- C-ABI of libmpcgpu.so: the MI355X-native batched NMPC solver.
 *
 * This is the drop-in boundary for the reference's solver plugin.  The reference loads a generated
 * OpEn/PyO3 module and calls (paths relative to /root/reference/):
 *
 *     built_solver = __import__(config.optimizer_name)          src/mpc_traj_tracker/trajectory_generator.py:70
 *     self.solver  = built_solver.solver()                      trajectory_generator.py:71
 *     solution     = self.solver.run(parameters, initial_guess) trajectory_generator.py:318
 *     solution.solution / .cost / .exit_status / .solve_time_ms trajectory_generator.py:320-323
 *     (stub signature incl. initial_lagrange_multipliers, initial_penalty: trajectory_generator.py:25-27)
 *
 * and the problem that solver was generated from is defined at
 *     src/mpc_traj_tracker/mpc/mpc_generator.py:160-297   (cost, constraints, solver settings)
 *     src/pkg_motion_model/motion_model.py:142-164         (unicycle RK4)
 *
 * Entry points below replace exactly that: create (= "build" the solver for a config), solve a batch of
 * independent parameter vectors `p` (same layout as mpc_generator.py:179-188), destroy.  Plain pointers
 * and sizes only; the library never frees caller memory, never throws; return 0 = ok, < 0 = error
 * (text via mpcgpu_last_error).  One handle per (device, stream); a handle is not thread-safe.
 * There is NO CPU fallback: without a usable HIP device mpcgpu_create fails.
 */
#ifndef MPCGPU_H
#define MPCGPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPCGPU_ABI_VERSION 8

/* replaces: the yaml config consumed by MpcModule.build (mpc_generator.py:151-158, config/mpc_default.yaml:7-55)
 * plus the SolverConfiguration of mpc_generator.py:285-293 (opengen defaults quoted there). */
typedef struct mpcgpu_config {
    int32_t N;        /* N_hor (1..64) */
    int32_t nu;       /* 2 */
    int32_t ns;       /* 3 */
    int32_t Nother;   /* max other robots        (<= 16) */
    int32_t Nstcobs;  /* max static obstacles    (<= 16) */
    int32_t nstcobs;  /* 12 = 4 edges x (b,a0,a1) */
    int32_t Ndynobs;  /* max dynamic obstacles   (1..32) */
    int32_t ndynobs;  /* 6 = (x,y,rx,ry,angle,alpha) */
    double ts;
    double lin_vel_min, lin_vel_max, ang_vel_max;
    double lin_acc_min, lin_acc_max, ang_acc_max;
    double vehicle_width, social_margin, fleet_weight; /* fleet_weight = 1000 (mpc_generator.py:216) */
    double tol;            /* 1e-4  */
    double delta_tol;      /* 1e-4  */
    double init_tol;       /* 1e-4  */
    double init_penalty;   /* 10    (mpc_generator.py:286) */
    double penalty_update; /* 5     */
    double tol_update;     /* 0.1   */
    double suff_decrease;  /* 0.1   */
    int32_t max_inner;     /* 500   */
    int32_t max_outer;     /* 10    */
    int32_t lbfgs_mem;     /* 10 (1..16) */
    int32_t device;        /* HIP device ordinal (>= 0) */
    double max_duration_us; /* 5e6 (mpc_generator.py:22); <= 0 disables the in-kernel clock test */
} mpcgpu_config;

/* exit_status codes; names as in config/mpc_default.yaml:54 */
enum { MPCGPU_CONVERGED = 0, MPCGPU_NOT_CONVERGED_ITERATIONS = 1, MPCGPU_NOT_CONVERGED_OUT_OF_TIME = 2,
       /* OpEn reports this case as an error (SolverError::NotFiniteComputation -> the binding returns None) */
       MPCGPU_NOT_FINITE_COMPUTATION = 3,
       /* not an OpEn status: the problem has more active rows than mpcgpu_reserve_shape promised */
       MPCGPU_SHAPE_EXCEEDED = 4 };

/* replaces: MpcModule.build + __import__(optimizer_name).solver()  (trajectory_generator.py:63-71) */
int32_t mpcgpu_create(const mpcgpu_config* cfg, void** handle);
void mpcgpu_destroy(void* handle);

/* error text of the last failing call on this handle (handle == NULL: last mpcgpu_create failure) */
const char* mpcgpu_last_error(void* handle);

/* len(p) for this config: mpc_generator.py:179-188 */
int32_t mpcgpu_num_params(void* handle);
int32_t mpcgpu_abi_version(void);

/*
 * replaces: solver.run(p, initial_guess, initial_lagrange_multipliers, initial_penalty) for B problems
 * (trajectory_generator.py:318).  HOST pointers.  u0 / y0 / c0 may be NULL (zeros / zeros / init_penalty:
 * every reference call site passes initial_guess=None, src/interface_mpc.py:82).  Output pointers other
 * than u, cost, status may be NULL.
 *   p      [B x np]    u0 [B x 2N]   y0 [B x 2N]   c0 [B]
 *   u      [B x 2N]    cost [B] (= f(u), penalty terms excluded)   status [B]
 *   inner_it, outer_it [B]   fpr [B] (last inner ||gamma*fpr||)   f2norm [B]   y_out [B x 2N]
 *   ms     [B]  device residency time of each solve (the solve_time_ms field of the plugin result)
 * Synchronous: results are in place on return.
 */
int32_t mpcgpu_solve_batch(void* handle, int32_t B, const double* p, const double* u0, const double* y0,
                           const double* c0, double* u, double* cost, int32_t* status, int32_t* inner_it,
                           int32_t* outer_it, double* fpr, double* f2norm, double* y_out, double* ms);

/* `stream` arguments are raw hipStream_t values and are used as given: NULL is HIP's null stream (which is also
 * torch's default stream, so a launch on it is ordered with the torch work around it).  This sentinel selects the
 * handle's own non-blocking stream instead (no ordering with any other stream: synchronise it yourself). */
#define MPCGPU_STREAM_OWN ((void*)(intptr_t)-1)

/* Same, DEVICE pointers (e.g. torch tensors' data_ptr()), enqueued on `stream` (see MPCGPU_STREAM_OWN).  Returns after
 * enqueueing the solve kernel (synchronise the stream before reading results).  The LDS carve of the launch is sized
 * from the batch's active-row maxima: without a reservation (mpcgpu_reserve_shape) this call contains one small
 * device->host read of those counts and blocks until the compaction kernel has run; with a reservation nothing is
 * read back and the call never blocks.  It can then be captured into a hipGraph, provided nothing has to be allocated or
 * opted into inside the capture: size the library's buffers first (mpcgpu_reserve_batch, or one eager call with at least
 * this batch size) -- a call that would have to grow a buffer or raise a kernel's LDS limit while `stream` is being
 * captured fails with -6 and a message instead of breaking the capture. */
int32_t mpcgpu_solve_batch_dev(void* handle, int32_t B, const double* p, const double* u0, const double* y0,
                               const double* c0, double* u, double* cost, int32_t* status, int32_t* inner_it,
                               int32_t* outer_it, double* fpr, double* f2norm, double* y_out, double* ms,
                               void* stream);

/*
 * Test hook -- replaces the generated CasADi functions cost(u, xi, p) / grad_cost / mapping_f1 / mapping_f2
 * that OpEn calls (built by mpc_generator.py:295-297): evaluates them once per problem on the GPU through
 * the very same device code the solver kernel uses.  HOST pointers.
 *   u [B x 2N]   xi [B x (1+2N)] = (c, y)   p [B x np]
 *   psi [B]  f [B]  grad [B x 2N]  F1 [B x 2N]  F2 [B x Ndynobs]     (any output may be NULL)
 */
int32_t mpcgpu_cost_grad_batch(void* handle, int32_t B, const double* u, const double* xi, const double* p,
                               double* psi, double* f, double* grad, double* F1, double* F2);

/* Device time (ms, HIP events on the launch stream) of the last solve call: parameter-compaction kernel and
 * solve kernel.  Valid after the stream has been synchronised. */
int32_t mpcgpu_last_timing(void* handle, double* prep_ms, double* solve_ms);

/* (ABI 7) The solve time of mpcgpu_last_timing split at the tail promotion (MPCGPU_OPT_TAIL_PROMOTION): main_ms = the throughput
 * (or latency) kernel including the ordering kernels in front of it, tail_ms = the continuation launch of the latency kernel
 * behind it (0 when none was enqueued).  main_ms + tail_ms = solve_ms. */
int32_t mpcgpu_last_tail_timing(void* handle, double* main_ms, double* tail_ms);

/* Work counters of the last solve call, per problem: psi evaluations executed and how many of them also produced
 * grad psi (the counts OpEn's generated `cost` / `grad_cost` functions would see, minus the redundant re-evaluation of
 * psi(u) in the Lipschitz update; the latency kernel reports the counts of the SEQUENTIAL algorithm, not its speculative
 * evaluations).  Waits for the last solve (the event recorded behind it; and for `stream`, if it is another one).  When the last
 * solve was CAPTURED into a hipGraph there is no such event: pass the stream the graph is launched on -- that stream, and no
 * other, is synchronised (never the whole device); calling this inside a capture fails with -6.  MPCGPU_STREAM_OWN after a solve
 * that was captured on a stream of the caller's cannot name the right stream: the whole device is drained then (which fails with
 * -6 while any stream of the device is being captured) rather than stale counters returned.  HOST output pointers. */
int32_t mpcgpu_last_eval_counts(void* handle, int32_t B, int32_t* n_psi, int32_t* n_grad, void* stream);

/* Batch-wide maxima of active entries seen by the last solve / cost_grad call: static obstacles, fleet
 * entries, dynamic-obstacle entries (sizes the LDS carve), and the LDS bytes per wavefront used.  After a call that ran the
 * latency kernel (mpcgpu_last_latency_kernel) or under mpcgpu_reserve_shape nothing was read back: the values are the bounds
 * the carve was sized for (configured maxima / reserved shape) and the LDS bytes of the whole workgroup. */
int32_t mpcgpu_last_shape(void* handle, int32_t* max_static, int32_t* max_fleet, int32_t* max_dyn,
                          int32_t* lds_bytes);

/*
 * Promise upper bounds on the ACTIVE (non-zero) static-obstacle, other-robot and dynamic-obstacle rows of every problem
 * of the following mpcgpu_solve_batch_dev calls, and what is known about the dynamic rows: var_shape = 1: a row may change
 * (rx, ry, angle, alpha) over the horizon (general tables); 0: every row keeps them (compact tables); 2: every row keeps them
 * AND is axis-aligned (angle 0 -- what the reference's own prediction feeder produces, src/main.py:77-85: the kernel without
 * the rotation into the ellipse frame, the same bits).  The LDS carve is then taken from these bounds instead of a read-back (single-robot callers
 * of the reference reserve the configured maxima: a lone wavefront does not care about the size of its carve).
 * A problem that exceeds the reservation is not solved: status = MPCGPU_SHAPE_EXCEEDED, cost = NaN, u = 0.
 * All three bounds negative = drop the reservation.  Results do not depend on the carve (bitwise).
 */
int32_t mpcgpu_reserve_shape(void* handle, int32_t max_static, int32_t max_fleet, int32_t max_dyn, int32_t var_shape);

/* Size the library-owned device buffers (workspace records, counters) for batches of up to B problems now, so that later
 * mpcgpu_solve_batch_dev calls allocate nothing (required before a call is captured into a hipGraph; growth is otherwise
 * automatic and drains the device first).  For a batch of the latency range (MPCGPU_OPT_TEAM_BATCH: by default up to two problems per compute unit) without a
 * reservation it also opts the one-launch latency kernel into its LDS size: inside a capture that whole range takes the
 * one-launch form (tables for the configured maxima, nothing read back). */
int32_t mpcgpu_reserve_batch(void* handle, int32_t B);

/*
 * Solver options that are not part of the reference's yaml / SolverConfiguration surface.  They select between readings
 * of the OpEn algorithm that cannot be checked against an OpEn build here; the oracle has the same
 * switch (oracle/mpc_oracle.h: ls_fallback).
 *   MPCGPU_OPT_LINESEARCH_FALLBACK  what follows 10 line-search halvings without acceptance:
 *       0 (default)  the last trial point (tau = 2^-10) becomes the iterate -- the effective behaviour of the published
 *                    PANOC engine, whose `tau = 0; u <- u_half` fallback is overwritten by the copy of u_plus into u
 *       1            tau = 0: the point u - gamma*fpr is evaluated and taken
 *   MPCGPU_OPT_PENALTY_STALL  (ABI 8) when the outer loop KEEPS the penalty c instead of multiplying it by penalty_update (the
 *       "penalty stall criterion" of the ALM / PM loop; mpc_generator.py:285-293 configures the solver this rule belongs to):
 *       0 (default)  "either": in the first outer iteration, or when ||y+ - y|| shrank by suff_decrease OR ||F2|| did -- the
 *                    published engine's `is_penalty_stall_criterion` as recalled (iteration == 0 || (n1 > 0 && dy+ <= theta dy + eps)
 *                    || (n2 > 0 && ||F2+|| <= theta ||F2|| + eps)).  While the acceleration constraints are inactive y+ = y = 0, the
 *                    first test holds and the penalty stays at init_penalty.
 *       1            "both": only when both shrank (the default of ABI <= 7 builds).
 *     Converged answers of the two readings are identical where neither ever raises the penalty and differ by up to ~5e-3 in u
 *     elsewhere (profiles/r06_stall_rule.txt); the oracle has the same switch (oracle/mpc_oracle.h: stall_rule).
 *   MPCGPU_OPT_PAIRING  problems per wavefront of the solve kernel:
 *       -1 (default) automatic: the faster layout as measured on the MI355X -- today one problem per wavefront for every
 *                    horizon
 *        0           one problem per wavefront (the layout BASELINE.json's north_star words)
 *        1           two problems per wavefront (rows 0-1 / rows 2-3; compiled for N_hor = 20, error for other horizons)
 *     The choice never depends on the batch.  Both layouts run the same source (csrc/mpc_kernels.hpp, lane models Solo / Duo);
 *     their results differ by floating-point summation order only.
 *   MPCGPU_OPT_TEAM_BATCH  largest batch that is solved by the LATENCY kernel (csrc/mpc_team.hpp: one problem per workgroup of
 *       four wavefronts that evaluate the Lipschitz test and the line-search trials of a PANOC step side by side; compaction
 *       fused; LDS carve from the configured maxima, nothing read back before the launch).  -1 (default, ABI 8): 2 x the number of
 *       compute units = the four-wavefront form; a larger value (up to 4 x the number of compute units, the default of ABI <= 7) adds the
 *       mid-range form with two wavefronts per problem (profiles/r06_team_sweep.txt: under the default penalty-stall reading the
 *       throughput kernel with its tail promotion is the faster one from 640 problems up); 0 switches it off.  Every horizon
 *       whose carve fits the 160 KiB of a compute unit; results are bitwise those of the throughput kernel.
 *   MPCGPU_OPT_ORDER  (ABI 5) in which order the throughput kernel starts the problems of a batch that is larger than what is
 *       resident at once (12 - 16 problems per compute unit, by the registers and the LDS carve of the kernel the batch runs on).
 *       A solve takes 10^1 .. 10^4 PANOC steps and whatever is long and starts last finishes on a draining GPU.
 *       1 (default)  longest first, by the psi-evaluation counts the PREVIOUS solve call of the same batch size left on this
 *                    handle (mpcgpu_last_eval_counts): in a receding-horizon loop problem i of this call is robot i one tick
 *                    later.  First call, or another batch size: as given.  Three small kernels on the launch stream (counting
 *                    sort on 1024 bins), no host synchronisation, capturable.
 *       0            as given (workgroup g solves problem g).
 *     Every problem is solved independently and writes the outputs of ITS index: results are bitwise the same in any order.
 *     The reference has no counterpart (one robot per solver.run call).
 *   MPCGPU_OPT_LINEAR_TABLES  (ABI 6) EXPERIMENT, effective only in the variant build libmpcgpu_linear40.so (-DMPC_LINEAR40=1; the
 *       product build accepts and ignores it).  1 (default): at N_hor = 40, with shape-constant axis-aligned dynamic rows (table
 *       kind 2) and a batch of more than 12 problems per compute unit, the centres of the dynamic rows travel as LINEAR TABLES: a
 *       constant-velocity prediction (src/main.py:77-85) lies on a straight line up to rounding, centre k = fma(n_k, u, fma(d, k,
 *       c0)) with a 16-bit integer n_k per coordinate and step -- lossless, checked per problem by the compaction kernel; 1.7
 *       instead of 5 KB of LDS per problem, so that a 128-register build of the solve kernel keeps 16 instead of 12 problems
 *       resident per compute unit; a problem with a row that does not fit is solved from the stored centres by a pick-up launch
 *       right behind.  Bitwise the same results -- and measured SLOWER (the 128-register build spills: profiles/
 *       r04_linear_tables_ab.txt), which is why the product does not carry it.  0: always the stored centres.
 *   MPCGPU_OPT_TAIL_PROMOTION  (ABI 7) the tail of a throughput launch.  A solve is one long dependency chain, so the last problems
 *       of a large batch finish on a draining GPU (0.07-0.08 s per launch whatever the batch).  -1 (default): once all but
 *       K problems of the launch have finished, every wavefront that is still running leaves at the START OF ITS NEXT INNER
 *       PROBLEM -- point, multipliers, penalty, tolerance and the outer loop's counters go into the problem's workspace record --
 *       and a continuation launch of the LATENCY kernel on the same stream (four wavefronts per problem: the evaluations of a step
 *       side by side: a cap-length solve in 34 ms against 55 ms for a lone wavefront of the throughput kernel) finishes those problems from exactly that point.  Automatic K:
 *       twice the teams that are resident at once (N_hor = 20: 4 x #CUs = 1024; N_hor = 40, one team per compute unit by its LDS
 *       carve: 2 x #CUs = 512 -- #CUs when the continuation is the launch BEHIND the throughput kernel, see
 *       MPCGPU_OPT_TAIL_CONCURRENT).  > 0: that K (beyond four times the residency: two wavefronts per problem); 0: off.  Same step functions
 *       on the same state: every output is BITWISE what the throughput kernel alone writes (tests/test_gpu_yield.py); nothing is
 *       read back, the call stays capturable.  The reference has no counterpart.
 *   MPCGPU_OPT_TAIL_POLL  (ABI 7) PANOC steps between two looks at the launch's finished-counter (a power of two, default 16;
 *       only the A/B build -DMPC_YIELD_STEP=1 looks inside an inner problem at all).
 *   MPCGPU_OPT_TAIL_WAVES  (ABI 7) wavefronts per promoted problem: 0 (default) by K as above, 2 or 4.  A TEST KNOB with a side effect:
 *       a non-zero value also fixes the team width of the ordinary latency-kernel path for batches between two and four problems
 *       per compute unit (mpcgpu_last_latency_kernel reports what ran).  Results do not depend on it (bitwise).
 *   MPCGPU_OPT_TAIL_CONCURRENT  (ABI 7) 1 (default): the continuation does not wait for the throughput launch to end -- it runs on a
 *       stream of the handle's own (lowest priority) WHILE that launch drains (behind a one-lane gate kernel that opens when the
 *       launch starts to promote and all its problems have begun; workgroup g waits for list entry g, every wait bounded by a
 *       wall-clock limit: 0.5 s there, 60 s at the gate), the launch stream waits for it at
 *       the end of the call, and a sweep launch behind both takes what might be left (normally nothing).  Same results, bit for
 *       bit; 247 -> 237 ms at B = 8192, 146 -> 133 ms at 4096, N_hor = 40 B = 4096: 295 -> 280 ms.  0: the continuation is the launch
 *       behind the throughput kernel -- always so while the call is being captured into a hipGraph, when K exceeds what the
 *       throughput kernel keeps resident, and while a launch of ANOTHER handle of this process is in flight on the device (seen by
 *       its end-of-call event; two handles that keep two streams busy: the other launch fills the drain, and workgroups of the
 *       latency kernel must not hold compute units that problems of a launch are waiting for).  That test is race-free between the
 *       THREADS of one process (a handle counts as in flight from the moment its solve call begins).  It cannot see other PROCESSES
 *       that share the GPU, graph replays of another handle, or foreign kernels of the caller on other streams: those cost time only
 *       (the lowest-priority side stream waits; every wait is bounded and a sweep launch behind the throughput kernel finishes what is
 *       left) -- tests/test_gpu_foreign_work.py -- and a process that shares its GPU that way sets this option to 0.
 *   MPCGPU_OPT_TAIL_GRADUAL  (ABI 8) EXPERIMENT, default 0 = off.  G > 0: with the concurrent continuation a problem may also leave the
 *       throughput launch -- at the start of an inner problem, once every problem of the launch has begun -- while (promoted + 1) * G <=
 *       finished and less than half of the list is taken, i.e. long before the last K problems.  Bitwise the same results
 *       (tests/test_gpu_yield.py) -- and measured SLOWER or equal on every batch and family (G = 4 / 8 / 16: up to +11 / +13 / +7 %,
 *       profiles/r06_tail_gradual_ab.txt): four wavefronts per promoted problem cost 2.5 x the issue slots of one for 1.6 x its speed,
 *       which pays only on a draining GPU.
 */
enum { MPCGPU_OPT_LINESEARCH_FALLBACK = 1, MPCGPU_OPT_PAIRING = 2, MPCGPU_OPT_TEAM_BATCH = 3, MPCGPU_OPT_ORDER = 4,
       MPCGPU_OPT_LINEAR_TABLES = 5, MPCGPU_OPT_TAIL_PROMOTION = 6, MPCGPU_OPT_TAIL_POLL = 7, MPCGPU_OPT_TAIL_WAVES = 8,
       MPCGPU_OPT_TAIL_CONCURRENT = 9, MPCGPU_OPT_PENALTY_STALL = 10, MPCGPU_OPT_TAIL_GRADUAL = 11 };
int32_t mpcgpu_set_option(void* handle, int32_t option, double value);

/* Register-allocation variant the last solve call was launched with: 3 (148 / 168 VGPRs) or 4 wavefronts per SIMD
 * (128 VGPRs; chosen when the LDS carve fits 16 times into a CU and the batch exceeds 12 problems per CU: N_hor = 20).
 * Both give bitwise identical results. */
int32_t mpcgpu_last_waves_per_simd(void* handle);

/* Wavefronts per problem of the latency kernel when the last solve call ran it (MPCGPU_OPT_TEAM_BATCH): 4, or 2 for batches
 * between two and four problems per compute unit (when MPCGPU_OPT_TEAM_BATCH admits them: the default range ends at two); 0 = the throughput
 * kernel ran. */
int32_t mpcgpu_last_latency_kernel(void* handle);

/* Dynamic-obstacle tables of the last solve / cost_grad launch, in the coding of mpcgpu_reserve_shape's var_shape: 1 general
 * (some row changes shape over the horizon, or the latency kernel ran), 0 shape-constant, 2 shape-constant and axis-aligned,
 * 3 the launch ran with linear centre tables (variant build only, see MPCGPU_OPT_LINEAR_TABLES).
 * A caller that wants read-back-free launches of the same batches can reserve exactly what the automatic rule found. */
int32_t mpcgpu_last_table_kind(void* handle);

/* Problems per wavefront of the last solve / cost_grad launch: 1 or 2 (MPCGPU_OPT_PAIRING). */
int32_t mpcgpu_last_problems_per_wavefront(void* handle);

/* 1 when the last solve call started its problems longest first (MPCGPU_OPT_ORDER), 0 when in the order given. */
int32_t mpcgpu_last_ordered(void* handle);

/* (ABI 7) Tail promotion of the last solve call (MPCGPU_OPT_TAIL_PROMOTION): returns the capacity K of its continuation launch
 * (0: none was enqueued -- latency-kernel batch, option off, two problems per wavefront).  `promoted` (may be NULL): how many
 * problems actually moved to the latency kernel; reading it waits for `stream` (the stream the call was enqueued on). */
int32_t mpcgpu_last_tail_promotion(void* handle, int32_t* promoted, void* stream);

/* (ABI 8) Health of the concurrent continuation of the last solve call (MPCGPU_OPT_TAIL_CONCURRENT): returns 1 when that call ran its
 * continuation beside the draining launch, 0 when behind it (or none).  `timeouts`: how many of its bounded waits ended by their
 * wall-clock limit instead of by their condition (the gate's 60 s, a workgroup's 0.5 s for its list entry) -- 0 in a healthy run;
 * a non-zero count means the side stream was starved (foreign work on the device) and the sweep launch finished those problems:
 * the results are the same, the call took longer.  Reading it waits for `stream` like mpcgpu_last_tail_promotion. */
int32_t mpcgpu_last_tail_timeouts(void* handle, int32_t* timeouts, void* stream);

/* ------------------------------------------------------------------------------------------------------------------------
 * Batched tracker harness on the device.  Replaces, for B robots per call and without a
 * host round trip, what the reference does per robot in Python around every solver.run:
 *     get_local_ref_traj            src/mpc_traj_tracker/trajectory_generator.py:206-232   (mpcgpu_tracker_window_dev)
 *     check_termination_condition   trajectory_generator.py:156-162                     \
 *     run_step: speed rule, parameter assembly   trajectory_generator.py:251-275         |  (mpcgpu_tracker_step_dev)
 *     run_solver                    trajectory_generator.py:318-323                      |
 *     run_step: taken states, predicted states   trajectory_generator.py:325-339        /
 * The parameter vector of mpc_generator.py:179-188 is never materialised: the compaction code of the solver reads the same
 * parameter indices straight from the tracker's arrays, so the solver sees bitwise the record it would have built from the
 * assembled vector.  All pointers are DEVICE memory owned by the caller (e.g. torch tensors), float64 / int32 / uint8.
 */
typedef struct mpcgpu_tracker {
    int32_t B;                 /* robots */
    int32_t ref_cap;           /* rows of every robot's slice of `ref` */
    int32_t action_steps;      /* inputs applied per tick (config action_steps) */
    int32_t _pad;
    double* states;            /* [B][3]  (x, y, theta), in/out */
    const double* goals;       /* [B][3]  final goals */
    double* last_actions;      /* [B][2]  in/out */
    const double* ref;         /* [B][ref_cap][3] global reference trajectories (get_global_ref_traj), padded */
    const int32_t* ref_len;    /* [B] */
    int32_t* idx_ref;          /* [B] in/out */
    const double* stc;         /* [B][Nstcobs * 12]      static half-plane rows (update_static_constraints) */
    const double* dyn;         /* [B][Ndynobs * 6 * N]   dynamic rows (update_dynamic_constraints) */
    const double* other;       /* [B][3 * N * Nother] other robots' predictions, or NULL = zeros */
    double* pred_states;       /* [B][N][3] out */
    uint8_t* active;           /* [B] in/out: 0 once the termination test has fired */
    double tuning[10];         /* tuning_params of the work mode (trajectory_generator.py:129-134) */
    double base_speed;         /* of the work mode */
    double low_speed;          /* config low_speed */
    double stc_weight, dyn_weight; /* obstacle weights (trajectory_generator.py:59) */
} mpcgpu_tracker;

/* refs_out [B][N][3]: every robot's local reference window; idx_ref is advanced like the reference's call does. */
int32_t mpcgpu_tracker_window_dev(void* handle, const mpcgpu_tracker* t, double* refs_out, void* stream);

/* One control tick: termination test, assembly (compact record, no padded vector), solve, taken / predicted states.
 * refs [B][N][3]: the reference every robot tracks this tick (get_action(current_ref_traj), src/interface_mpc.py:82-88).
 * u0 [B][2N] or NULL (cold start, what the reference does).  Outputs as mpcgpu_solve_batch_dev; actions_out [B][2] (may be
 * NULL) = the applied first input, 0 for robots that are done.  Needs mpcgpu_reserve_shape (no count read-back: the call only
 * enqueues) and follows the throughput / latency kernel rule of mpcgpu_solve_batch_dev. */
int32_t mpcgpu_tracker_step_dev(void* handle, const mpcgpu_tracker* t, const double* refs, const double* u0, double* u,
                                double* cost, int32_t* status, int32_t* inner_it, int32_t* outer_it, double* actions_out,
                                void* stream);

/* The DQN's proposal for B robots: agent [B][agent_stride] rows starting (x, y, theta, v, w); action [B] (int64, 0..8);
 * rl_ref [B][steps][2] (src/pkg_dqn/environment/agent.py:86-145, src/main.py:193-202).  limits[8] = acc_max, acc_min,
 * angacc_max, angacc_min, speed_min, speed_max, angvel_min, angvel_max. */
int32_t mpcgpu_rl_reference_dev(void* handle, int32_t B, const double* agent, int32_t agent_stride, const int64_t* action,
                                double ts, int32_t steps, double ref_speed, const double* limits, double* rl_ref, void* stream);

/* HintSwitcher.switch (src/main_pre.py:27-52) for B robots + the reference each tracks this tick: polygons [B][O][V][2] (rings
 * padded by repeating the last vertex), valid [B][O], states [B][3], original [B][N][3], rl_ref [B][rl_steps][2], live [B] or
 * NULL; switch_on [B] (uint8) / detach_cnt [B] (int32) in/out; chosen [B][N][3] out (the proposal with the original heading
 * column where the switch is on -- ref_traj_filter with decay 1, src/main.py:34-41 -- else the original). */
int32_t mpcgpu_hint_switch_dev(void* handle, int32_t B, int32_t N, int32_t O, int32_t V, const double* polygons,
                               const uint8_t* valid, const double* states, const double* original, const double* rl_ref,
                               int32_t rl_steps, const uint8_t* live, double switch_distance, double detach_distance,
                               double detach_steps, uint8_t* switch_on, int32_t* detach_cnt, double* chosen, void* stream);

/* Test hook: copy the library's workspace records of the last B problems to the host (B x mpcgpu_workspace_stride doubles;
 * the first mpcgpu_workspace_record doubles of each are the compact problem record, the rest is solver scratch). */
int32_t mpcgpu_debug_read_workspace(void* handle, int32_t B, double* out);
int32_t mpcgpu_workspace_stride(void* handle);
int32_t mpcgpu_workspace_record(void* handle);
/* Test hooks: run only the compaction kernel on B host parameter vectors / only the tracker's assembly kernel (both fill the
 * workspace records and return when they are complete). */
int32_t mpcgpu_debug_prep(void* handle, int32_t B, const double* p);
int32_t mpcgpu_debug_tracker_assemble(void* handle, const mpcgpu_tracker* t, const double* refs);
/* Test hook (ABI 6; compiled horizons N_hor = 20 / 40): the L-BFGS operator of PANOC alone.  U, R [B][m + 1][2N]: a recorded
 * sequence of iterates u_j and residuals gamma*fpr_j; they are fed to the buffer the way the solver feeds it (pairs
 * s = u_j - u_{j-1}, y = r_j - r_{j-1}, C-BFGS acceptance, memory 10) and d = H r_m comes back twice: evaluated in the Gram form the
 * product kernels use (d_gram) and by the two-loop recursion of the `lbfgs` crate (d_twoloop), [B][2N] each; pairs [B][2] =
 * accepted pairs in the buffer of either form (may be NULL).  HOST pointers. */
int32_t mpcgpu_debug_lbfgs_direction(void* handle, int32_t B, int32_t m, const double* U, const double* R, double* d_gram,
                                     double* d_twoloop, int32_t* pairs);

#ifdef __cplusplus
}
#endif
#endif /* MPCGPU_H */
