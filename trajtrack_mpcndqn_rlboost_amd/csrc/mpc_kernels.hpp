// mpc_kernels.hpp -- device code of the batched NMPC solver for gfx950 (MI355X, wave64).
//
// One problem instance per 64-lane wavefront (one wavefront per workgroup):
//   * "vector lanes"  k < N hold step k of every horizon vector: (v_k, w_k) of u, grad, FPR, L-BFGS work
//     vectors, multipliers ... -- 2 doubles per lane, so n = 2N-dim vector algebra is one instruction and an
//     inner product is one DPP wave reduction;
//   * "item lanes"    all 64 lanes: (k, sub) = (lane % N, lane / N) split the (step x object) stage-cost
//     terms of step k (reference-path segments, static polygons, dynamic ellipses, fleet discs);
//   * the compacted problem tables and the L-BFGS memory live in LDS; HBM is touched once per solve.
//
// What is restated (paths relative to /root/reference/):
//   cost / constraints : src/mpc_traj_tracker/mpc/mpc_generator.py:25-54,85-130,160-272
//   unicycle RK4       : src/pkg_motion_model/motion_model.py:142-164 (closed form: Simpson on the heading)
//   solver             : the OpEn algorithm the reference generates at mpc_generator.py:269-297
//                        (PANOC + L-BFGS + ALM/PM; published algorithm)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

// Wave priority (s_setprio; round 4): the phases of an evaluation that are chains of dependent cross-lane steps (rollout scans, adjoint,
// and the solver logic between two evaluations) run at a raised priority, the item loops -- independent work per lane, the part that
// fills the issue slots other wavefronts leave -- at the base priority: a wavefront in a chain issues as soon as its operand arrives
// instead of waiting its turn.  No instruction but the two s_setprio per evaluation; same bits.  N_hor = 20 only: -1 % kernel time at
// B = 32768, -2.5 % at B = 8192; with the raised priority ending at the end of the evaluation (logic at base) the gain is lost.  At
// N_hor = 40 (three wavefronts per SIMD) it is -1 % at B = 16384 and +1.5 % at B = 4096, config 3's batch: not used there
// (profiles/r04_setprio_ab.txt).
#define MPC_PRIO_CHAIN() __builtin_amdgcn_s_setprio(2)
#define MPC_PRIO_ITEMS() __builtin_amdgcn_s_setprio(0)
#define MPC_ITEM_LOOP _Pragma("unroll 1")  // item loops: keep the loads of one iteration in flight, not of all (other policies: +-1 %, round 2)
// Round 6: item phase B (the weighted gradients of the hard dynamic-obstacle constraint) re-derived every (row, step) item of phase A --
// table reads, ellipse-frame coordinates, the indicator -- to form  W_i * d(Ih)/d(position).  With MPC_CACHE_B the unweighted derivative
// of the first MPC_CACHE_B_TRIPS trips (rows c_isub, c_isub + LPS, ...: nine rows at N_hor = 20) stays in registers from phase A (zero
// outside the ellipse) and phase B is one table read and two fused multiply-adds per trip; rows beyond them are re-derived as before.
// The same expression values in the same order: same bits.  Compiled horizons with a uniform lane split only (N_hor = 20).
#ifndef MPC_CACHE_B
#define MPC_CACHE_B 1
#endif
// ... the same for the balanced walk of N_hor = 40: the first MPC_CACHE_B40 (1..4) trips of a lane's OWN rows (the foreign items and later
// trips are re-derived).  An experiment knob: 0 unless measured faster (the 168-register kernel has no register to spare).
#ifndef MPC_CACHE_B40
#define MPC_CACHE_B40 0
#endif
// L-BFGS pair update: the sums of an accepted pair next to those of the acceptance test (see PanocLbfgsGram::update).  Round-6 experiment, measured:
// the latency kernel LOSES 3 % (27.3 -> 28.2 ms for a cap-length solve), the throughput kernel is indifferent: not used.
#ifndef MPC_LB_HOIST_SUMS
#define MPC_LB_HOIST_SUMS 0
#endif

namespace mpcgpu {

// Optional phase profiler (build with -DMPC_PROFILE; never in the shipped library): shader-clock cycles per
// phase, accumulated per wavefront and added to a device-global table at the end of the solve.
#ifdef MPC_PROFILE
constexpr int NPROF = 24;
__device__ unsigned long long g_prof[NPROF];
struct Prof {
    long long t[NPROF];
    long long last;
    __device__ void start() { for (int i = 0; i < NPROF; ++i) t[i] = 0; last = __builtin_readcyclecounter(); }
    __device__ void mark(int i) { const long long n = __builtin_readcyclecounter(); t[i] += n - last; last = n; }
    __device__ void count(int i) { t[i] += 1; }
    __device__ void flush() { if (threadIdx.x == 0) for (int i = 0; i < NPROF; ++i) atomicAdd(&g_prof[i], (unsigned long long)t[i]); }
};
#define PROF_ARG , Prof& prof
#define PROF_PASS , prof
#define PROF_MARK(i) prof.mark(i)
#define PROF_COUNT(i) prof.count(i)
#else
#define PROF_ARG
#define PROF_PASS
#define PROF_MARK(i)
#define PROF_COUNT(i)
#endif

// Row sums of the hinge matrix H[i][k] = max(0, inside_ellipse(row i, step k)) (the hard dynamic-obstacle constraint F2): every
// item lane adds its positive terms to D_i in LDS (ds_add_f64: the lanes of one instruction are served in a fixed order, so the
// sum is deterministic and the same in every kernel that runs this code).  Rounds 1-2 stored the matrix (Kd x N doubles of LDS)
// and lane i summed its row in a serial loop.
__device__ __forceinline__ void lds_add(double* p, double v) {
    __builtin_amdgcn_ds_atomic_fadd_f64((__attribute__((address_space(3))) double*)p, v);
}

// Linear centre tables + the four-wavefront build of the N_hor = 40 solve kernel (round 4): measured SLOWER than the product
// (profiles/r04_linear_tables_ab.txt: the 128-register build of that kernel spills 92 VGPRs), so it is compiled only into the
// variant build libmpcgpu_linear40.so (`make variants`, -DMPC_LINEAR40=1), which tests/test_gpu_linear_tables.py keeps bitwise equal.
#ifndef MPC_LINEAR40
#define MPC_LINEAR40 0
#endif
constexpr int WAVE = 64;
constexpr int PREP_STATIC_LDS = (32 + (MPC_LINEAR40 ? 64 : 0)) * 4;   // bytes of static LDS of prep_problem (s_entry, s_ue): part of every kernel that inlines it
constexpr int HDR = 64;        // header doubles per problem in the workspace
constexpr int SEGW = 9;        // LDS / workspace doubles per reference segment: s1x, s1y, dx, dy, 1/(|d|^2+1e-16), then the
                               // bounding circle (cx, cy, R) of ALL segments from this one to the last, + 1 pad: the odd
                               // stride spreads the records of neighbouring steps over all LDS banks (an even stride
                               // gave 3-way conflicts on every segment read: -1.8 % kernel time)
constexpr int STCW = 12;       // doubles per static obstacle    (b[4], a0[4], a1[4])
constexpr int DYNW = 9;        // workspace record per (dyn row, step): cx, cy, cosA, sinA, ihx, ihy, isx, isy, wgt
constexpr int DYNP = 2;        // shape-constant LDS record per (row, step): cx, cy
constexpr int DYNC = 7;        // shape-constant LDS record per row: cosA, sinA, ihx, ihy, isx, isy, alpha
                               // (the item weight q_dyn[k] * alpha is formed where it is used: q_dyn is one value per step)
constexpr int DYNL = 6;        // linear centre tables: per row x0, y0, dx, dy, ux, uy (centre of step k = fma(n_k, u, fma(d, k, c0)), n_k a 16-bit integer)
constexpr int PARTW = 5;       // doubles per item-lane partial  (gx, gy, best, bgx, bgy)
constexpr int MAX_MEM = 16;
// Reference segments per item lane that are evaluated unconditionally before the suffix-circle test takes over.  Rounds 1-3: 2
// (the 6 nearest segments of a step at N_hor = 20; the pruned loop then ran in 0.1 % of the evaluations).  1 = the 3 nearest:
// the pruned loop runs in 2.2 % of the evaluations and one unconditional trip per evaluation is gone: -3.0 % kernel time at
// N_hor = 20, -1.6 % at N_hor = 40, same bits (the same segments win in the same order; profiles/archive/r03_step_loop_ab.txt).
constexpr int SEG_WIN = 1;

// header slots (doubles); 0..17 are p[0..17] of the reference layout
enum { H_X0 = 0, H_Y0 = 1, H_TH0 = 2, H_XG = 3, H_YG = 4, H_THG = 5, H_VINIT = 6, H_WINIT = 7, H_QVEL = 9, H_RV = 11, H_RW = 12,
       H_QN = 13, H_QTHN = 14, H_QRPD = 15, H_ACC = 16, H_WACC = 17 };
enum { H_KS = 18, H_KF = 19, H_KD = 20, H_CTH0 = 21, H_STH0 = 22, H_NPF = 23, H_NPD = 24, H_VAR = 25 /* 1: some dynamic row changes shape over the horizon */,
       H_ENTRY = 26 /* .. +Ndynobs (<= 32) */, H_ROT = 60 /* 1: some active dynamic row is rotated (angle != 0) */,
       H_NLIN = 61 /* 1: some active dynamic row does not move on a straight line (see LINEAR CENTRE TABLES) */ };
// batch-wide reductions written by the compaction kernel
enum { CNT_KS = 0, CNT_KF = 1, CNT_KD = 2, CNT_VARSHAPE = 3, CNT_ROTATED = 4 /* some active dynamic row is not an axis-aligned ellipse (angle != 0) */,
       CNT_NONLINEAR = 5 /* some active dynamic row is not a straight-line prediction */,
       CNT_NL_COUNT = 6 /* length of the list of such problems (nl_list_kernel) */,
       CNT_FINISHED = 8 /* problems of the running throughput launch that have written their results (tail promotion, see YIELD) */,
       CNT_YIELDED = 9 /* length of the list of problems that left the throughput launch for the latency kernel */,
       CNT_LISTED = 10 /* entries of that list that are complete (record + list slot written): FINISHED + LISTED = every problem decided */,
       CNT_STARTED = 11 /* problems of the running throughput launch that have begun (gate of the concurrent continuation) */,
       CNT_TIMEOUTS = 12 /* waits of the concurrent continuation that ended by their wall-clock limit (gate: 60 s; a workgroup waiting for its
                            list entry: 0.5 s) instead of by their condition -- mpcgpu_last_tail_timeouts; the sweep launch finishes such entries */,
       CNT_WORDS = 14 };

struct KParams {
    int N, Nother, Nstcobs, Ndynobs, np, mem;
    int max_inner, max_outer;
    int ls_fallback;  // line search without acceptance after 10 halvings: 0 = keep the last trial point, 1 = tau = 0
    int stall_rule;   // when the penalty is kept: 0 = EITHER infeasibility shrank by theta (the published engine), 1 = BOTH did
    double ts, inv_ts;
    double vmin, vmax, wmax, amin, amax, aamax;
    double W2, social, fleetw;
    double tol, delta_tol, init_tol, init_penalty, penalty_update, tol_update, suff_decrease;
    long long max_ticks;  // wall_clock64 ticks (100 MHz); <= 0 disables
    // parameter-vector offsets (mpc_generator.py:179-188)
    int r0, c0, os0, od0, qs0, qd0;
    // workspace (global) layout per problem, doubles
    int ws_stride, ws_vref, ws_seg, ws_stc, ws_fxy, ws_dyn, ws_qd, ws_alpha, ws_dynl, ws_dynr, ws_lbs, ws_lby, ws_lold;
    // LDS layout (doubles), strides use the batch maxima mKs/mKf/mKd
    int mKs, mKf, mKd;
    int reserved;  // 1: the carve comes from mpcgpu_reserve_shape, problems are checked against it on the device
    int l_seg, l_stc, l_fxy, l_dyn, l_dync, l_dynl, l_qd, l_pos, l_H, l_W, l_part, l_bal, l_stash, l_hd, l_S, l_Y, l_rho, l_alpha, l_old, l_gg, l_total;
    int l_wstride, l_xch;  // team kernel (mpc_team.hpp): doubles per wavefront work block, offset of the exchange area
    // tail promotion (YIELD below): a problem of the throughput launch leaves at the start of an inner problem once `yield_from`
    // problems of the launch have finished; 0 = off.  yield_cap: capacity of the list, ws_yield: offset of the saved iteration state in
    // the workspace record; yield_mask (builds with -DMPC_YIELD_STEP=1 only): the counter is also polled when (step & yield_mask) == 0.
    int yield_from, yield_cap, yield_mask, ws_yield;
    // concurrent continuation (mpc_team.hpp, CONCURRENT): 1 = the workgroups of the latency kernel wait for their list entries while
    // the throughput launch is still running; yield_total = problems of that launch (the list is final once FINISHED + LISTED reach it)
    int yield_persist, yield_total;
    // gradual promotion (round 6, MPCGPU_OPT_TAIL_GRADUAL; only with the concurrent continuation): once every problem of the launch has
    // begun, a problem may ALSO leave at the start of an inner problem while (promoted + 1) * yield_grad <= finished and fewer than half of
    // the list is taken -- the teams fill the compute units the finished wavefronts free instead of waiting for the last yield_cap
    // problems.  0 = off.
    int yield_grad;
};

// ------------------------------------------------------------------------------------------------
// TAIL PROMOTION (round 5).  A solve is one long dependency chain, so the last problems of a throughput launch finish on a
// draining GPU: 0.07-0.08 s per launch whatever the batch (a quarter of a launch of 8192 problems).  The latency kernel
// (mpc_team.hpp) runs the SAME iteration 1.6 x faster than a wavefront that has its SIMD to itself (34 vs 55 ms for a cap-length solve; ~97 ms for
// one of four wavefronts on a full SIMD) -- when the GPU has room.  So: every problem of the throughput
// launch counts itself as finished (CNT_FINISHED); once all but `yield_cap` problems of the launch have finished, a wavefront that
// reaches the START OF AN INNER PROBLEM writes the state of its outer loop -- point, multipliers, penalty, tolerance, counters, the
// position of the L-BFGS ring: the PANOC cache and the buffer are empty there -- into its workspace record, appends its problem to a
// list and leaves; a continuation launch of the latency kernel (solve_kernel_team with io.ylist set, grid = yield_cap) starts that
// inner problem -- behind the throughput kernel on the same stream (workgroups beyond the device-side list length leave at once;
// the form a hipGraph records), or, by default, on a stream of the handle's own WHILE the throughput launch drains (mpc_team.hpp,
// CONCURRENT: the record is published with a release fence, the list entry after it).  Both kernels run the same step functions
// on the same state: every output is BITWISE what the throughput kernel alone would have written (tests/test_gpu_yield.py).
// Nothing is read back; capturable.
// Record at ws_yield (doubles): YS_* scalars, then [N][8] = (u, grad, half step, multipliers) of every step, then the L-BFGS
// scalars that live in LDS (rho, Gram matrices) -- gradient, half step and the L-BFGS part are written by MPC_YIELD_STEP builds only.
// ------------------------------------------------------------------------------------------------
// Where a problem may leave.  0 (product): at the start of an inner problem only -- the state machine of solve_body, ten times per
// solve; the loop of the PANOC steps is untouched (a running problem takes up to one inner problem, <= max_inner steps, to get
// there).  1 (A/B build): also at every yield_mask + 1-th PANOC step, with the whole PANOC cache and the L-BFGS buffer in the
// record -- measured: the extra exit of the step loop costs registers in the 128-VGPR build (64 -> 92 spilled VGPRs, +59 lane
// reads per step), more than the 2-5 ms the finer grain saves per launch.
#ifndef MPC_YIELD_STEP
#define MPC_YIELD_STEP 0
#endif
#ifndef MPC_DUP_SHARED
#define MPC_DUP_SHARED 0   // 1: measurement build, see eval_point
#endif
enum { YS_C = 0, YS_GAMMA, YS_IG, YS_LIP, YS_SIGMA, YS_COST, YS_GG, YS_D2H, YS_AKKT, YS_NFPR, YS_IP, YS_DYN, YS_F2N, YS_LASTFPR, YS_FFINAL,
       YS_HGAMMA, YS_ELAPSED, YS_ITER, YS_NUMITER, YS_FLAGS /* 1 cont_iters, 2 cont_time, 4 a step has completed, 8 L-BFGS buffer empty, 16 left at the start of an inner problem */,
       YS_ALMIT, YS_NOUTER, YS_INNERTOT, YS_STATUS, YS_NEVAL, YS_NEVALG, YS_LBACTIVE, YS_LBHEAD, YS_TRN, YS_TRPSI, YS_SCALARS = 32 };
constexpr int YS_VECW = 8;
__host__ __device__ constexpr int yield_even_c(int x) { return (x + 1) & ~1; }
// doubles of the LDS region `gg` (Gram matrices, or the alpha scratch of the two-loop form): see fixed_lds
__host__ __device__ constexpr int gg_doubles_c(int N, int mem, bool gram) {
    return gram ? yield_even_c(mem * mem + mem * (mem + 1) / 2) : yield_even_c(mem);
}

// How H * (gamma fpr) is evaluated: 1 = Gram form (PanocLbfgsGram, round 3), 0 = two-loop recursion (PanocLbfgs; the build
// `make variants` keeps as libmpcgpu_twoloop.so for A/B runs).  Two problems per wavefront (Duo) always take the two-loop form.
#ifndef MPC_LBFGS_GRAM
#define MPC_LBFGS_GRAM 1
#endif

// ------------------------------------------------------------------------------------------------
// LDS carve, fixed part.  Every region whose size depends only on (N_hor, L-BFGS memory) comes FIRST, at offsets that are
// compile-time constants for the kernels with a compiled horizon; the tables whose size follows the batch's active rows (static
// polygons, fleet discs, dynamic ellipses) come after them.  A kernel with a compiled horizon then addresses positions, stash,
// partials, header, segments, Gram matrices ... through the 16-bit immediate offset of the DS instructions: no base pointer in
// an SGPR (the scalar file of the solve kernel is over-subscribed: 94 spilled SGPRs, every reload a v_readlane on the VALU), no
// address arithmetic beyond the lane's own scaled index.  The host (mpcgpu.hip fill_lds_layout / fill_team_layout) fills
// KParams::l_* from the SAME function, so the generic kernel and the latency kernel read the identical layout at run time.
// ------------------------------------------------------------------------------------------------
struct FixedLds { int hd, seg, pos, stash, part, W, bal, rho, gg, S, Y, old, end; };
__host__ __device__ constexpr int even_c(int x) { return (x + 1) & ~1; }
// (N_hor = 40 as well since the stash-free 168-register kernel: see stash_stride_c)
__host__ __device__ constexpr bool gram_shape(int N, int mem) { return MPC_LBFGS_GRAM && (N == 20 || N == 40) && mem == 10; }
// Item-lane partials (eval_point): every item lane beyond the vector lanes parks PARTW = 5 doubles.  With a compiled horizon that
// nearly divides the wavefront (N_hor = 20: 60 item lanes) the split is uniform and the last lanes idle -- the rule of eval_point.
__host__ __device__ constexpr int part_doubles_c(int N, int mem) {
    const bool compiled = (N == 20 || N == 40) && mem == 10;
    const bool uniform = compiled && (64 % N) * 5 <= N;
    const int item_lanes = uniform ? (64 / N) * N : 64;
    return (item_lanes - N) * 5;
}
// The stash (6 doubles per step) and the positions before it are dead between two evaluations; the Gram form of the L-BFGS step
// uses them as scratch there: the operands of pass 1 ((r, y) pairs of every chunk slot) followed by the row coefficients.
// Doubles per step in the stash: the Simpson sums and their derivatives (Cx, Sy, dCw, dSw) and, where registers are short, the
// point itself (v, w).  The long compiled horizon runs a 168-register kernel that can carry (v, w) through the item phase, and
// its carve is what decides the residency there: LDS is handed out in 1280-byte granules, 13 440 B are 11 granules = 11
// wavefronts per CU -- measured: 2.58 resident wavefronts per SIMD on a 21-round batch (tools/occupancy_probe.sh), where 12 per
// CU would show 2.8 -- and 12 wavefronts need <= 12 800 B.  Without (v, w) (640 B) and with q_dyn in the pad double of the
// segment records (320 B) the N_hor = 40 carve is 12 480 B.
// With the Gram form at N_hor = 40 (its matrices: 1240 B) the four Simpson values stay in registers as well (stride 0): 11 200 B
// + 1160 B = 12 360 B, still 12 per CU.
// `minw`: wavefronts per SIMD the kernel is compiled for.  The 128-register build of the long horizon (minw = 4, round 4) cannot
// carry the four Simpson values through the item phase: it parks them (stride 4; (v, w) stay in registers -- with all six parked
// the carve of the benchmark shape would be 10 704 B, over the 10 240 B that sixteen wavefronts per compute unit allow).
// Balanced walk of the dynamic rows where the item lanes do not divide the steps evenly (N_hor = 40: steps 0-23 have two item
// lanes, steps 24-39 one): see eval_point.  0 = the row walk of rounds 1-3 (`make variants`: libmpcgpu_rowwalk40.so).
#ifndef MPC_BALANCED40
#define MPC_BALANCED40 1
#endif
__host__ __device__ constexpr bool balanced_shape(int N, int mem) { return MPC_BALANCED40 && N == 40 && mem == 10; }
// gradient accumulators of the single-lane steps (2 doubles each) that the helper lanes add their foreign items to
__host__ __device__ constexpr int bal_doubles_c(int N, int mem) { return balanced_shape(N, mem) ? 2 * (N - (64 - N)) : 0; }
#ifndef MPC_STW40_W4
#define MPC_STW40_W4 4   // experiment knob of round 4 (6: all six values parked, 14 wavefronts per compute unit)
#endif
__host__ __device__ constexpr int stash_stride_c(int N, int mem, int minw = 3) {
    return (N == 40 && mem == 10) ? (minw >= 4 ? MPC_STW40_W4 : (gram_shape(N, mem) ? 0 : 4)) : 6;
}
__host__ __device__ constexpr int stash_doubles_c(int N, int mem, int minw = 3) {
    int need = N * stash_stride_c(N, mem, minw);
    if (gram_shape(N, mem)) {
        const int R = 2 * mem, G = 32 / mem, CL = (N + G - 1) / G, G2 = 64 / N, CR = (R + G2 - 1) / G2;
        int scratch = even_c(G * CL * 4) + even_c(G2 * CR);      // pass-1 operands (r, y) + row coefficients
        const int p2 = (G2 - 1) * N * 2;                          // partials of pass 2 (they reuse the operand area)
        if (p2 > scratch) scratch = p2;
        // the scratch starts at the positions and runs through the stash into the region of the item partials / hinge sums
        // that follows it: all three are dead between two evaluations
        const int after = part_doubles_c(N, mem) > 2 * 32 ? part_doubles_c(N, mem) : 2 * 32;
        const int room = N * 2 + (stash_stride_c(N, mem, minw) == 0 ? after : 0);
        if (scratch - room > need) need = scratch - room;
    }
    return need;
}
constexpr int HW_ROWS = 32;   // doubles reserved for the hinge row sums D_i; the weights W_i follow at this offset (Ndynobs <= 32)
__host__ __device__ constexpr FixedLds fixed_lds(int N, int mem, bool lbfgs_in_lds, int minw = 3) {
    FixedLds f{};
    int o = 0;
    f.hd = o; o += 64;
    f.seg = o; o += even_c(N * 9);
    f.pos = o; o += N * 2;
    f.stash = o; o += stash_doubles_c(N, mem, minw);
    // hinge row sums + weights and the item-lane partials are never live together (LDS operations of one wavefront execute in
    // order): one region
    f.part = o; f.W = o + HW_ROWS;
    { const int ps = part_doubles_c(N, mem); o += ps > 2 * HW_ROWS ? ps : 2 * HW_ROWS; }
    f.bal = o; o += bal_doubles_c(N, mem);
    f.rho = o; o += even_c(mem);
    f.gg = o;   // Gram matrices s_i.y_j (full) + y_i.y_j (packed symmetric), or the alpha scratch of the two-loop form
    o += gram_shape(N, mem) ? even_c(mem * mem + mem * (mem + 1) / 2) : even_c(mem);
    f.S = f.Y = f.old = o;
    if (lbfgs_in_lds) {
        f.S = o; o += mem * N * 2;
        f.Y = o; o += mem * N * 2 + N * 2;  // + the zero row
        f.old = o; o += N * 4;
    }
    f.end = o;
    return f;
}

// compile-time L-BFGS memory of the kernels with a compile-time horizon (the launcher sends other memories to the generic kernel)
template <int NT> struct MemOf { static constexpr int value = NT ? 10 : 0; };
// compile-time horizon NT (0 = runtime horizon from KParams; DPP row counts then cover the whole wave)
template <int NT>
struct Dim {
    static constexpr int ROWS_V = NT ? (NT + 15) / 16 : 4;  // rows holding vector lanes
    static constexpr int ROWS_I = 4;                        // every lane is an item lane
};

struct BatchPtrs {
    const double* p; const double* u0; const double* y0; const double* c0;
    double* u; double* cost; int32_t* status; int32_t* inner_it; int32_t* outer_it;
    double* fpr; double* f2norm; double* y_out; double* ms;
    double* ws; int* counts;
    int32_t* evals;  // [B][2] psi evaluations / of those with gradient (library-owned; read by mpcgpu_last_eval_counts)
    const int32_t* perm;  // throughput kernel: workgroup g solves problem perm[g] (NULL: problem g) -- MPCGPU_OPT_ORDER, mpc_order.hpp
    const int* nsel;      // != NULL: only the first *nsel entries of `perm` are problems of this launch (device-side count: the
                          // launch that picks up the problems a linear-table launch left out)
    int32_t* ylist;       // tail promotion: problems that left the throughput launch for the latency kernel (length: counts[CNT_YIELDED])
    double* trace;   // -DMPC_TRACE builds only: [B][trace_cap][TRACE_W] decision trace, one record per PANOC step
    int trace_cap;
};
constexpr int TRACE_W = 12;  // outer, step, c, L, gamma, ||gamma fpr||, psi(u), Lipschitz doublings, L-BFGS pairs, halvings, tau, psi(u+)

// ------------------------------------------------------------------------------------------------
// wave primitives on DPP (data-parallel primitives: cross-lane operands inside VALU instructions, no LDS
// round trip).  gfx950 is a GFX9-family wave64 target: row_shr / row_shl / row_bcast / wave_shr are legal.
// A double moves as two 32-bit DPP movs.  Sums are bit-identical in every lane (they come out of SGPRs).
// ------------------------------------------------------------------------------------------------
enum : int {
    DPP_ROW_SHL0 = 0x100, DPP_ROW_SHR0 = 0x110, DPP_WAVE_SHL1 = 0x130, DPP_WAVE_SHR1 = 0x138,
    DPP_BCAST15 = 0x142, DPP_BCAST31 = 0x143
};
// lanes whose source is outside the row / wave, or that are masked off, receive 0
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ double dpp0(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, true);
    return __hiloint2double(hi, lo);
}
// ... receive `fill` instead (identity element of a multiplicative scan)
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ double dppf(double x, double fill) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(__double2loint(fill), lo, CTRL, ROW_MASK, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(__double2hiint(fill), hi, CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_d(double x, int l) {  // uniform broadcast of lane l
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double uniform(double x) {  // tell the compiler x is wave-uniform
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(x));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(x));
    return __hiloint2double(hi, lo);
}
// One wavefront per workgroup: LDS operations of a wave execute in program order, so cross-lane exchange
// through LDS needs no s_barrier -- only the compiler must not reorder across the exchange point.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// inclusive prefix sum inside every 16-lane row
__device__ __forceinline__ double row_prefix(double x) {
    x += dpp0<DPP_ROW_SHR0 + 1>(x);
    x += dpp0<DPP_ROW_SHR0 + 2>(x);
    x += dpp0<DPP_ROW_SHR0 + 4>(x);
    x += dpp0<DPP_ROW_SHR0 + 8>(x);
    return x;
}
__device__ __forceinline__ double row_suffix(double x) {
    x += dpp0<DPP_ROW_SHL0 + 1>(x);
    x += dpp0<DPP_ROW_SHL0 + 2>(x);
    x += dpp0<DPP_ROW_SHL0 + 4>(x);
    x += dpp0<DPP_ROW_SHL0 + 8>(x);
    return x;
}
// sum over the first ROWS*16 lanes (lanes that do not take part must carry 0); uniform result.
// Row totals are chained with row_bcast15 / row_bcast31 (2 DPP moves + 1 add each; rows without a source read 0 through
// bound_ctrl, their lanes are not used) and the total is read from the last lane: 17 / 20 VALU instructions for 2 / 4 rows
// instead of 20 / 27 with one v_readlane pair per row.
// (The same totals on the matrix pipe -- two v_mfma_f64_4x4x4 with matrices of ones + two DPP steps -- were measured in rounds 1
// and 3: no gain on the chain-bound kernel, +1.5 % time on the issue-bound one: an f64 MFMA holds the issue port for its four
// passes.  DESIGN.md section 7; the experiment's code is in the history, commit 802f579.)
template <int ROWS>
__device__ __forceinline__ double wave_sum_u(double x) {
    x = row_prefix(x);  // lane 15 of each row = row total
    if (ROWS == 1) return readlane_d(x, 15);
    if (ROWS == 3) {    // rows 1 and 3 only, then rows 2-3: lane 47 = r2 + (r1 + r0)
        x += dpp0<DPP_BCAST15, 0xA>(x);
        x += dpp0<DPP_BCAST31, 0xC>(x);
        return readlane_d(x, 47);
    }
    x += dpp0<DPP_BCAST15>(x);                 // lane 31 = r1 + r0, lane 63 = r3 + r2 (the other lanes are not used)
    if (ROWS == 4) x += dpp0<DPP_BCAST31>(x);  // lane 63 = (r3 + r2) + (r1 + r0)
    return readlane_d(x, 16 * ROWS - 1);
}
// two sums over lanes 0..31 for the price of one: b travels in rows 2-3 (v_permlane32_swap), one DPP sequence serves both.
// Bitwise the same totals as wave_sum_u<2>(a), wave_sum_u<2>(b).
__device__ __forceinline__ void wave_sum2_u(double a, double b, double& sa, double& sb) {
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    double x = __hiloint2double((int)hi[0], (int)lo[0]);  // lanes 0..31: a, lanes 32..63: b (lanes 0..31 of it)
    x = row_prefix(x);
    x += dpp0<DPP_BCAST15>(x);
    sa = readlane_d(x, 31);
    sb = readlane_d(x, 63);
}
// inclusive prefix sum over lanes 0..16*ROWS-1 (lanes >= n must carry 0)
template <int ROWS>
__device__ __forceinline__ double scan_prefix(double x) {
    x = row_prefix(x);
    if (ROWS > 1) x += dpp0<DPP_BCAST15, 0xA>(x);  // last lane of rows 0/2 -> rows 1/3
    if (ROWS > 2) x += dpp0<DPP_BCAST31, 0xC>(x);  // lane 31 -> rows 2,3
    return x;
}
// inclusive suffix sum over lanes 0..16*ROWS-1 (lanes >= n must carry 0)
template <int ROWS>
__device__ __forceinline__ double scan_suffix(double x, int lane) {
    x = row_suffix(x);  // first lane of each row = row total
    // cascade from the top row down: each row adds the (already completed) first lane of the row above it
    if (ROWS > 3) { const double t = readlane_d(x, 48); if (lane >= 32 && lane < 48) x += t; }
    if (ROWS > 2) { const double t = readlane_d(x, 32); if (lane >= 16 && lane < 32) x += t; }
    if (ROWS > 1) { const double t = readlane_d(x, 16); if (lane < 16) x += t; }
    return x;
}
__device__ __forceinline__ double shift_up1(double x, int lane, double first) {  // lane k gets lane k-1
    const double t = dpp0<DPP_WAVE_SHR1>(x);
    return lane == 0 ? first : t;
}
__device__ __forceinline__ double shift_down1(double x) {  // lane k gets lane k+1 (lane 63: 0)
    return dpp0<DPP_WAVE_SHL1>(x);
}
__device__ __forceinline__ double clampd(double x, double lo, double hi) { return fmin(fmax(x, lo), hi); }
// The same clamp for WAVE-UNIFORM bounds (kernel arguments), as the two instructions it is.  fmin / fmax make the compiler
// canonicalise every operand it did not compute itself (v_max_f64 x, x, x -- also for the bounds, on every call): 10
// instructions for the two clamps of a half step instead of 4.  v_max_f64 / v_min_f64 return the other operand for a quiet NaN
// exactly as fmax / fmin do; the bits are those of clampd.
__device__ __forceinline__ double clamp_u(double x, double lo, double hi) {
    double r;
    asm("v_max_f64 %0, %1, %2\n\tv_min_f64 %0, %0, %3" : "=&v"(r) : "v"(x), "s"(lo), "s"(hi));
    return r;
}
// A kernel argument / a zero that has to be formed WHERE IT IS USED.  Without this the compiler forms loop invariants such as
// 0.5 * ts, 2 * fleet weight or a plain 0.0 once, before the solver loop, keeps them in VGPRs for the whole solve -- and, the
// 128-VGPR build being full, spills them: the reload (a scratch load + s_waitcnt vmcnt(0)) then sits at the head of every
// evaluation's dependency chain.  One VALU instruction at the point of use is cheaper than that by two orders of magnitude.
__device__ __forceinline__ double here_s(double x) { asm volatile("" : "+s"(x)); return x; }   // stays an SGPR pair
__device__ __forceinline__ double zero_here() { double z = 0.0; asm volatile("" : "+v"(z)); return z; }

// inclusive prefix PRODUCT of unit complex numbers (re, im) over lanes 0..16*ROWS-1 (lanes >= n carry 1+0i)
// x moved by DPP, lanes without a source (or in rows that are masked off) receive 1.0.  The low word of 1.0 is zero: where every
// row is written it travels in the zero-filling form and only the high word needs its fill value preloaded.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_one(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    if (ROW_MASK == 0xf) lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, true);
    else lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0x3ff00000, hi, CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void cmul_step(double& re, double& im) {
    // identity element 1 + 0i for lanes without a source: the real part needs its fill value preloaded, the imaginary part is the
    // zero-filling form (two v_mov less per step)
    const double pr = dpp_one<CTRL, ROW_MASK>(re), pi = dpp0<CTRL, ROW_MASK>(im);
    const double nr = re * pr - im * pi, ni = re * pi + im * pr;
    re = nr; im = ni;
}
template <int ROWS>
__device__ __forceinline__ void scan_cprod(double& re, double& im) {
    cmul_step<DPP_ROW_SHR0 + 1, 0xf>(re, im);
    cmul_step<DPP_ROW_SHR0 + 2, 0xf>(re, im);
    cmul_step<DPP_ROW_SHR0 + 4, 0xf>(re, im);
    cmul_step<DPP_ROW_SHR0 + 8, 0xf>(re, im);
    if (ROWS > 1) cmul_step<DPP_BCAST15, 0xA>(re, im);
    if (ROWS > 2) cmul_step<DPP_BCAST31, 0xC>(re, im);
}
// ------------------------------------------------------------------------------------------------
// Lane models of the solver (template parameter P of load_problem / eval_point / solve_body).
//   Solo<NT>: one problem per wavefront.  Iteration scalars are wave-uniform and live in SGPRs (sums come back through
//             v_readlane), every lane is an item lane, N <= 64.
//   Duo<NT> : TWO problems per wavefront, rows 0-1 = problem 2*blockIdx, rows 2-3 = problem 2*blockIdx+1 (N <= 31).  At
//             N = 20 the scans, the PANOC vector algebra and the ~40 inner products per iteration keep 20 of 64 lanes busy
//             in the Solo layout; here the same instruction serves both problems.  The two halves run their own state
//             machines (the compiler's EXEC masking does the bookkeeping: a "scalar" of the iteration is a VGPR whose 32
//             lanes hold bitwise the same value), every cross-lane operation stays inside a half:
//               * all-reduce: xor butterfly inside each 16-lane row (quad_perm, row_half_mirror, row_mirror: addition is
//                 commutative, so every lane of a row ends with the same bits), then v_permlane16_swap exchanges the two
//                 rows of each half;
//               * prefix scans: row_shr + row_bcast15 with row mask 0xA (rows 0 -> 1 and 2 -> 3 in one instruction);
//               * suffix scans: row_shl, then the first lane of the upper row via v_permlane16_swap + row_newbcast:0.
//             (tools/probes/duo_lanes.hip checks these on the hardware, also with one half masked off.)
// ------------------------------------------------------------------------------------------------
enum : int { DPP_QUAD_XOR1 = 0xB1, DPP_QUAD_XOR2 = 0x4E, DPP_ROW_MIRROR = 0x140, DPP_ROW_HALF_MIRROR = 0x141, DPP_ROW_NEWBCAST0 = 0x150 };
// total of every 16-lane row, bitwise identical in all of its lanes
__device__ __forceinline__ double row_allsum(double x) {
    x += dpp0<DPP_QUAD_XOR1>(x);
    x += dpp0<DPP_QUAD_XOR2>(x);
    x += dpp0<DPP_ROW_HALF_MIRROR>(x);
    x += dpp0<DPP_ROW_MIRROR>(x);
    return x;
}
// lane for lane: x of the even row of this lane's half (rows 0 / 2) and of the odd row (rows 1 / 3)
__device__ __forceinline__ void row_pair(double x, double& even_row, double& odd_row) {
    const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    even_row = __hiloint2double((int)b[0], (int)a[0]);
    odd_row = __hiloint2double((int)b[1], (int)a[1]);
}
// total of every 32-lane half, bitwise identical in all of its lanes
__device__ __forceinline__ double half_allsum(double x) {
    x = row_allsum(x);
    double e, o;
    row_pair(x, e, o);
    return e + o;
}
// inclusive suffix sum inside every 32-lane half
__device__ __forceinline__ double half_suffix(double x) {
    x = row_suffix(x);
    double e, o;
    row_pair(x, e, o);
    return x + dpp0<DPP_ROW_NEWBCAST0, 0x5>(o);  // rows 0 / 2 += first lane of rows 1 / 3
}

template <int NT>
struct Solo {
    static constexpr bool DUO = false;
    static constexpr int W = WAVE;             // lanes per problem
    static constexpr int RV = Dim<NT>::ROWS_V;  // rows holding vector lanes
    static constexpr int RI = 4;               // rows holding item lanes
    static __device__ __forceinline__ int lane() { return threadIdx.x; }
    static __device__ __forceinline__ int half() { return 0; }
    static __device__ __forceinline__ int problem() { return blockIdx.x; }
    static __device__ __forceinline__ double uni(double x) { return uniform(x); }
    static __device__ __forceinline__ unsigned uni_u(unsigned x) { return (unsigned)__builtin_amdgcn_readfirstlane((int)x); }
    template <int ROWS> static __device__ __forceinline__ double sum(double x) { return wave_sum_u<ROWS>(x); }
    // two independent sums over the vector lanes
    static __device__ __forceinline__ void sum2(double a, double b, double& sa, double& sb) {
        if (RV <= 2) wave_sum2_u(a, b, sa, sb);
        else { sa = wave_sum_u<RV>(a); sb = wave_sum_u<RV>(b); }
    }
    static __device__ __forceinline__ bool any(bool c) { return __ballot(c) != 0ull; }
    template <int ROWS> static __device__ __forceinline__ double prefix(double x) { return scan_prefix<ROWS>(x); }
    template <int ROWS> static __device__ __forceinline__ void cprod(double& re, double& im) { scan_cprod<ROWS>(re, im); }
    template <int ROWS> static __device__ __forceinline__ double suffix(double x, int lane) { return scan_suffix<ROWS>(x, lane); }
    // two inclusive suffix sums over the vector lanes for the price of one when they fit half a wavefront (b travels in rows 2-3,
    // like prefix2): the same additions in the same order per element, bitwise the two separate scans
    template <int ROWS> static __device__ __forceinline__ void suffix2(double a, double b, int lane, double& sa, double& sb) {
        if (ROWS <= 2) {
            auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
            auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
            double x = __hiloint2double((int)hi[0], (int)lo[0]);   // lanes 0..31: a, lanes 32..63: b (its lanes 0..31)
            x = row_suffix(x);
            if (ROWS > 1) {   // rows 0 / 2 add the (completed) first lane of rows 1 / 3
                const double t1 = readlane_d(x, 16), t3 = readlane_d(x, 48);
                if ((lane & 31) < 16) x += (lane < 32 ? t1 : t3);
            }
            lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(x), (unsigned)__double2loint(x), false, false);
            hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(x), false, false);
            sa = x;
            sb = __hiloint2double((int)hi[1], (int)lo[1]);
        } else { sa = scan_suffix<ROWS>(a, lane); sb = scan_suffix<ROWS>(b, lane); }
    }
    static __device__ __forceinline__ double from_lane(double x, int l) { return readlane_d(x, l); }
    // two inclusive prefix sums over the vector lanes for the price of one when they fit half a wavefront: b travels in rows 2-3
    // (v_permlane32_swap), one DPP sequence serves both halves, b comes back the same way.  Bitwise the two separate scans.
    template <int ROWS> static __device__ __forceinline__ void prefix2(double a, double b, double& pa, double& pb) {
        if (ROWS <= 2) {
            auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
            auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
            double x = __hiloint2double((int)hi[0], (int)lo[0]);   // lanes 0..31: a, lanes 32..63: b (its lanes 0..31)
            x = row_prefix(x);
            if (ROWS > 1) x += dpp0<DPP_BCAST15, 0xA>(x);          // rows 0 -> 1 and 2 -> 3
            lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(x), (unsigned)__double2loint(x), false, false);
            hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(x), false, false);
            pa = x;                                                // lanes 0..31 (the others are not vector lanes)
            pb = __hiloint2double((int)hi[1], (int)lo[1]);         // lanes 0..31 receive what lanes 32..63 hold
        } else { pa = scan_prefix<ROWS>(a); pb = scan_prefix<ROWS>(b); }
    }
};
template <int NT>
struct Duo {
    static_assert(NT >= 2 && NT <= 31, "two problems per wavefront need a compile-time horizon of at most 31 steps");
    static constexpr bool DUO = true;
    static constexpr int W = WAVE / 2;
    static constexpr int RV = 2;
    static constexpr int RI = 2;
    static __device__ __forceinline__ int lane() { return threadIdx.x & 31; }
    static __device__ __forceinline__ int half() { return threadIdx.x >> 5; }
    static __device__ __forceinline__ int problem() { return 2 * blockIdx.x + (threadIdx.x >> 5); }
    static __device__ __forceinline__ double uni(double x) { return x; }
    static __device__ __forceinline__ unsigned uni_u(unsigned x) { return x; }
    template <int ROWS> static __device__ __forceinline__ double sum(double x) { return half_allsum(x); }
    static __device__ __forceinline__ void sum2(double a, double b, double& sa, double& sb) { sa = half_allsum(a); sb = half_allsum(b); }
    static __device__ __forceinline__ bool any(bool c) {
        const unsigned long long m = __ballot(c);
        return ((threadIdx.x & 32) ? (unsigned)(m >> 32) : (unsigned)m) != 0u;
    }
    template <int ROWS> static __device__ __forceinline__ double prefix(double x) { return scan_prefix<2>(x); }
    template <int ROWS> static __device__ __forceinline__ void cprod(double& re, double& im) { scan_cprod<2>(re, im); }
    template <int ROWS> static __device__ __forceinline__ double suffix(double x, int) { return half_suffix(x); }
    template <int ROWS> static __device__ __forceinline__ void suffix2(double a, double b, int, double& sa, double& sb) { sa = half_suffix(a); sb = half_suffix(b); }
    static __device__ __forceinline__ double from_lane(double x, int l) { return half_allsum(lane() == l ? x : 0.0); }
    template <int ROWS> static __device__ __forceinline__ void prefix2(double a, double b, double& pa, double& pb) {
        pa = scan_prefix<2>(a); pb = scan_prefix<2>(b);
    }
};

// Double-precision literals cannot be instruction operands on gfx950; the compiler materialises each one in a VGPR
// pair and keeps it there for the whole solver loop (two registers per literal, >20 literals on the hot path).
// Reading them from constant memory makes them scalar loads into SGPRs instead.
__constant__ double KTAB[27] = {
    -1.66666666666666324348e-01, 8.33333333332248946124e-03, -1.98412698298579493134e-04,   // sin: S1..S6
    2.75573137070700676789e-06, -2.50507602534068634195e-08, 1.58969099521155010221e-10,
    4.16666666666666019037e-02, -1.38888888888741095749e-03, 2.48015872894767294178e-05,    // cos: C1..C6
    -2.75573143513906633035e-07, 2.08757232129817482790e-09, -1.13596475577881948265e-11,
    1.0 / 6.0,                 // 12
    1.0 + 1e-9,                // 13: safety factor of the pruning radii
    1.0 + 1e-6,                // 14: safety factor of the estimated sqrt(best) (sqrt_upper)
    0.78,                      // 15: |half-step heading increment| up to which the polynomials are used
    1.0 / (1e-6 * 1e-6),       // 16: inverse squared semi-axis of a zero-padded dynamic row: 1/((0+1e-6)^2)
    0.95,                      // 17: gamma = 0.95 / L                               [OpEn PANOC constants from here]
    1e-12,                     // 18: delta of the Lipschitz-estimate perturbation
    1e-6,                      // 19: epsilon of the Lipschitz-estimate perturbation; also the Lipschitz-update slack
    1e9,                       // 20: largest Lipschitz estimate
    1e-10,                     // 21: smallest Lipschitz estimate; also the s'y acceptance threshold of the L-BFGS buffer
    2.220446049250313e-16,     // 22: machine epsilon used in the ALM comparisons
    2.2250738585072014e-308,   // 23: smallest normal double (||s||^2 test of the L-BFGS buffer)
    1e-8,                      // 24: C-BFGS epsilon
    1e12,                      // 25: bound of the multiplier set Y
    (1.0 - 0.95) / 4.0};       // 26: sigma = (1 - 0.95) / (4 gamma); the division by 4 is exact
enum { K_SIXTH = 12, K_REACH = 13, K_SQRT = 14, K_SMALL = 15, K_IPAD = 16, K_GAMMA_L = 17, K_DELTA_LIP = 18, K_EPS_LIP = 19,
       K_MAX_LIP = 20, K_MIN_L = 21, K_EPS = 22, K_DBLMIN = 23, K_CBFGS = 24, K_YBOUND = 25, K_SIGMA = 26 };

// Upper bound of sqrt(x) for the pruning radii: the hardware estimate (v_rsq_f64, ~2^-24 relative) inflated by 1e-6 instead of
// the ~20-instruction correctly rounded square root.  Pruning only needs a radius that is not too SMALL; results do not change.
__device__ __forceinline__ double sqrt_upper(double x, double inflate) {
    return x > 0.0 ? x * __builtin_amdgcn_rsq(x) * inflate : 0.0;
}

// sin / cos on [-pi/4, pi/4] (fdlibm kernel polynomials, error < 1 ulp there)
__device__ __forceinline__ void sincos_small(double x, const double* k, double& s, double& c) {
    const double z = x * x;
    const double ps = k[0] + z * (k[1] + z * (k[2] + z * (k[3] + z * (k[4] + z * k[5]))));
    s = x + x * z * ps;
    const double pc = k[6] + z * (k[7] + z * (k[8] + z * (k[9] + z * (k[10] + z * k[11]))));
    c = 1.0 - 0.5 * z + z * z * pc;
}

// ------------------------------------------------------------------------------------------------
// parameter compaction: p (mpc_generator.py:179-188 layout, mostly zero padding) -> per-problem tables.
// Only ACTIVE rows get a table entry.  Zero padding keeps its reference semantics analytically:
//   * all-zero static obstacles contribute exactly 0 (prod of max(0,0)^2)           -> dropped
//   * all-zero other-robot rows are `npf` identical discs at the origin             -> closed form in the kernel
//   * all-zero dynamic-obstacle rows are `npd` identical ellipses (rx=ry=0, alpha=0) at the origin -> closed form
// one 64-thread block per problem.
// ------------------------------------------------------------------------------------------------
// One wavefront compacts problem b (lane = its lane index 0..63).  `counts` may be NULL (no batch maxima wanted).
// The parameter vector is read through `p[i]` of a SOURCE type: ParamVector = the reference's padded vector in memory (the
// plugin boundary); mpc_tracker.hpp adds a source that maps the same indices onto the device-resident tracker state of a
// robot, so that a tick of the batched tracker writes this compact record directly -- same code, same arithmetic, same bits.
struct ParamVector {
    const double* p;
    __device__ __forceinline__ double operator[](int i) const { return p[i]; }
};
template <class Src>
__device__ __forceinline__ void prep_problem(const KParams& kp, const Src& p, double* __restrict__ ws,
                                             int* counts, int lane) {
    const int N = kp.N;

    if (lane < 18) ws[lane] = p[lane];
    if (lane == 0) {
        double s, c;
        sincos(p[2], &s, &c);
        ws[H_CTH0] = c;
        ws[H_STH0] = s;
    }
    // reference segments i = 0..N-1: (ref_i, ref_{i+1}), ref_N := ref_{N-1}  (mpc_generator.py:194-195)
    for (int i = lane; i < N; i += WAVE) {
        const int i2 = (i + 1 < N) ? i + 1 : N - 1;
        const double s1x = p[kp.r0 + 3 * i], s1y = p[kp.r0 + 3 * i + 1];
        const double dx = p[kp.r0 + 3 * i2] - s1x, dy = p[kp.r0 + 3 * i2 + 1] - s1y;
        double* sg = ws + kp.ws_seg + SEGW * i;
        sg[0] = s1x; sg[1] = s1y; sg[2] = dx; sg[3] = dy;
        sg[4] = 1.0 / (dx * dx + dy * dy + 1e-16);
        // bounding circle of the reference points i..N-1 (every remaining segment lies inside it)
        double xlo = s1x, xhi = s1x, ylo = s1y, yhi = s1y;
        for (int j = i + 1; j < N; ++j) {
            const double qx = p[kp.r0 + 3 * j], qy = p[kp.r0 + 3 * j + 1];
            xlo = fmin(xlo, qx); xhi = fmax(xhi, qx); ylo = fmin(ylo, qy); yhi = fmax(yhi, qy);
        }
        const double bcx = 0.5 * (xlo + xhi), bcy = 0.5 * (ylo + yhi);
        double r2 = 0.0;
        for (int j = i; j < N; ++j) {
            const double ex = p[kp.r0 + 3 * j] - bcx, ey = p[kp.r0 + 3 * j + 1] - bcy;
            r2 = fmax(r2, ex * ex + ey * ey);
        }
        sg[5] = bcx; sg[6] = bcy; sg[7] = sqrt(r2) * (1.0 + 1e-12) + 1e-300; sg[8] = 0.0;
        ws[kp.ws_vref + i] = p[kp.r0 + 3 * N + i];
        ws[kp.ws_qd + i] = p[kp.qd0 + i];
    }
    // ---- static obstacles: lane o checks obstacle o
    int Ks;
    {
        bool nz = false;
        if (lane < kp.Nstcobs)
            for (int e = 0; e < STCW; ++e) nz |= (p[kp.os0 + STCW * lane + e] != 0.0);
        const unsigned long long m = __ballot(nz);
        Ks = __popcll(m);
        if (nz) {
            const int idx = __popcll(m & ((1ull << lane) - 1ull));
            for (int e = 0; e < STCW; ++e) ws[kp.ws_stc + STCW * idx + e] = p[kp.os0 + STCW * lane + e];
        }
    }
    // ---- other robots: lane j checks robot j (x,y only; theta never enters the cost)
    int Kf;
    {
        bool nz = false;
        if (lane < kp.Nother)
            for (int k = 0; k < N; ++k) {
                const int o = kp.c0 + lane * 3 * N + 3 * k;
                nz |= (p[o] != 0.0) || (p[o + 1] != 0.0);
            }
        const unsigned long long m = __ballot(nz);
        Kf = __popcll(m);
        if (nz) {
            const int idx = __popcll(m & ((1ull << lane) - 1ull));
            for (int k = 0; k < N; ++k) {
                const int o = kp.c0 + lane * 3 * N + 3 * k;
                ws[kp.ws_fxy + (idx * N + k) * 2] = p[o];
                ws[kp.ws_fxy + (idx * N + k) * 2 + 1] = p[o + 1];
            }
        }
        if (lane == 0) ws[H_NPF] = (double)(kp.Nother - Kf);
    }
    // ---- dynamic obstacles: lane i checks row i
    int Kd;
    bool varshape = false, rotated = false, nonlinear = false;
    __shared__ int s_entry[32];  // original row -> entry (or -1 for padded rows); Ndynobs <= 32
    {
        bool nz = false;
        if (lane < kp.Ndynobs) {
            const int q0 = kp.od0 + lane * 6 * N;
            auto q = [&](int t) { return p[q0 + t]; };
            for (int t = 0; t < 6 * N; ++t) nz |= (q(t) != 0.0);
            for (int k = 1; k < N; ++k)  // semi-axes, angle and alpha constant over the horizon?
                varshape |= (q(6 * k + 2) != q(2)) || (q(6 * k + 3) != q(3)) || (q(6 * k + 4) != q(4)) || (q(6 * k + 5) != q(5));
            for (int k = 0; k < N; ++k) rotated |= nz && (q(6 * k + 4) != 0.0);  // angle 0 <=> cos = 1, sin = 0 exactly
        }
        const unsigned long long m = __ballot(nz);
        Kd = __popcll(m);
        const int idx = nz ? __popcll(m & ((1ull << lane) - 1ull)) : -1;
        if (lane < kp.Ndynobs) {
            ws[H_ENTRY + lane] = (double)idx;
            s_entry[lane] = idx;
        }
        if (lane == 0) ws[H_NPD] = (double)(kp.Ndynobs - Kd);
        wave_sync();  // s_entry is exchanged inside ONE wavefront
        // (row, step) items: sincos of the ellipse angle and the inverse squared semi-axes, once per solve
        for (int t = lane; t < kp.Ndynobs * N; t += WAVE) {
            const int i = t / N, k = t - i * N;
            const int e = s_entry[i];
            if (e < 0) continue;
            const int q0 = kp.od0 + i * 6 * N + 6 * k;
            double* d = ws + kp.ws_dyn + (e * N + k) * DYNW;
            double sa, ca;
            sincos(p[q0 + 4], &sa, &ca);
            const double rx = p[q0 + 2], ry = p[q0 + 3];
            d[0] = p[q0]; d[1] = p[q0 + 1]; d[2] = ca; d[3] = sa;
            d[4] = 1.0 / ((rx + 1e-6) * (rx + 1e-6));
            d[5] = 1.0 / ((ry + 1e-6) * (ry + 1e-6));
            d[6] = 1.0 / ((rx + kp.social + 1e-6) * (rx + kp.social + 1e-6));
            d[7] = 1.0 / ((ry + kp.social + 1e-6) * (ry + kp.social + 1e-6));
            d[8] = p[kp.qd0 + k] * p[q0 + 5];  // q_dyn[k] * alpha
            if (k == 0) ws[kp.ws_alpha + e] = p[q0 + 5];
        }
#if MPC_LINEAR40
        // LINEAR CENTRE TABLES (round 4).  A constant-velocity prediction (est_dyn_obs_positions, src/main.py:77-85) puts the
        // centres of a row on a straight line up to rounding.  With c0 = the first centre and d = (last - first) / (N - 1) the
        // predictor lin_k = fma(d, k, c0) misses the stored centre by a few units of the operands' last place; every such
        // residual is an integer multiple n_k of u = the smallest ulp among the row's centres and predictor values, and
        // fma(n_k, u, lin_k) gives the stored double back EXACTLY (the exact sum is that double).  Per row 6 doubles + one 16-bit
        // integer per coordinate and step instead of 16 bytes per step: 5.0 -> 1.7 KB of LDS at N_hor = 40 with 8 rows --
        // lossless.  A row that does not fit 16 bits (a curved or scanner prediction, a coordinate closer than ~1/4000 of the
        // largest one to zero) flags the PROBLEM (H_NLIN): it is solved from the stored centres by a second launch.
        __shared__ int s_ue[2 * 32];   // per original row: smallest binary exponent among its x (y) centres and predictor values
        if (lane < kp.Ndynobs) { s_ue[2 * lane] = 4096; s_ue[2 * lane + 1] = 4096; }
        wave_sync();
        auto lin_of = [&](int i, int k, double& lx, double& ly, double& dxl, double& dyl, double& x0, double& y0) {
            const int qr = kp.od0 + i * 6 * N, ql = qr + 6 * (N - 1);
            x0 = p[qr]; y0 = p[qr + 1];
            const double inv = 1.0 / (double)(N - 1);
            dxl = (p[ql] - x0) * inv; dyl = (p[ql + 1] - y0) * inv;
            lx = __builtin_fma(dxl, (double)k, x0); ly = __builtin_fma(dyl, (double)k, y0);
        };
        auto expo = [](double z) -> int {   // binary exponent of a normal double; zero does not constrain the unit; anything else: unusable
            const int e = (int)((__double_as_longlong(z) >> 52) & 0x7ff);
            return z == 0.0 ? 4096 : (e == 0 || e == 0x7ff) ? -4096 : e - 1023;
        };
        for (int t = lane; t < kp.Ndynobs * N; t += WAVE) {
            const int i = t / N, k = t - i * N;
            if (s_entry[i] < 0) continue;
            double lx, ly, dxl, dyl, x0, y0;
            lin_of(i, k, lx, ly, dxl, dyl, x0, y0);
            const int q0 = kp.od0 + i * 6 * N + 6 * k;
            const int ex = min(expo(p[q0]), expo(lx)), ey = min(expo(p[q0 + 1]), expo(ly));
            atomicMin(&s_ue[2 * i], ex); atomicMin(&s_ue[2 * i + 1], ey);
        }
        wave_sync();
        for (int t = lane; t < kp.Ndynobs * N; t += WAVE) {
            const int i = t / N, k = t - i * N;
            const int e = s_entry[i];
            if (e < 0) continue;
            double lx, ly, dxl, dyl, x0, y0;
            lin_of(i, k, lx, ly, dxl, dyl, x0, y0);
            const int q0 = kp.od0 + i * 6 * N + 6 * k;
            // unit = 2^(exponent - 52); a row of zeros (or of unusable values) gets the unit 1 and fails below unless it is exact
            auto unit = [](int ex, double& u, double& iu) {
                const bool ok = ex > -900 && ex < 900;
                u = ok ? __longlong_as_double((long long)(ex - 52 + 1023) << 52) : 1.0;
                iu = ok ? __longlong_as_double((long long)(52 - ex + 1023) << 52) : 1.0;
            };
            double ux, iux, uy, iuy;
            unit(s_ue[2 * i], ux, iux); unit(s_ue[2 * i + 1], uy, iuy);
            const double nx = (p[q0] - lx) * iux, ny = (p[q0 + 1] - ly) * iuy;
            const bool fits = fabs(nx) <= 32767.0 && fabs(ny) <= 32767.0 && nx == (double)(int)nx && ny == (double)(int)ny &&
                              __builtin_fma(nx, ux, lx) == p[q0] && __builtin_fma(ny, uy, ly) == p[q0 + 1];
            nonlinear |= !fits;
            const int wx = fits ? (int)nx : 0, wy = fits ? (int)ny : 0;
            reinterpret_cast<unsigned*>(ws + kp.ws_dynr)[e * N + k] = ((unsigned)wx & 0xFFFFu) | ((unsigned)wy << 16);
            if (k == 0) { double* L = ws + kp.ws_dynl + e * DYNL; L[0] = x0; L[1] = y0; L[2] = dxl; L[3] = dyl; L[4] = ux; L[5] = uy; }
        }
#endif
    }
    const bool any_var = __ballot(varshape) != 0ull, any_rot = __ballot(rotated) != 0ull, any_nonlin = __ballot(nonlinear) != 0ull;
    if (lane == 0) {
        ws[H_KS] = (double)Ks; ws[H_KF] = (double)Kf; ws[H_KD] = (double)Kd; ws[H_VAR] = any_var ? 1.0 : 0.0;
        ws[H_ROT] = any_rot ? 1.0 : 0.0;
        ws[H_NLIN] = any_nonlin ? 1.0 : 0.0;
        if (counts) {
            atomicMax(counts + CNT_KS, Ks);
            atomicMax(counts + CNT_KF, Kf);
            atomicMax(counts + CNT_KD, Kd);
            if (any_var) atomicMax(counts + CNT_VARSHAPE, 1);
            if (any_rot) atomicMax(counts + CNT_ROTATED, 1);
            if (any_nonlin) atomicMax(counts + CNT_NONLINEAR, 1);
        }
    }
}
__global__ __launch_bounds__(WAVE) void prep_kernel(KParams kp, BatchPtrs io, int B) {
    const int b = blockIdx.x;
    if (b >= B) return;
    prep_problem(kp, ParamVector{io.p + (size_t)b * kp.np}, io.ws + (size_t)b * kp.ws_stride, io.counts, threadIdx.x);
}

// problems whose dynamic rows do not fit the linear centre tables (H_NLIN), as a list for the pick-up launch
__global__ __launch_bounds__(256) void nl_list_kernel(const double* __restrict__ ws, int ws_stride, int B, int* counts, int32_t* list) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B && ws[(size_t)b * ws_stride + H_NLIN] != 0.0) list[atomicAdd(counts + CNT_NL_COUNT, 1)] = b;
}

// ------------------------------------------------------------------------------------------------
// per-wave problem context
// ------------------------------------------------------------------------------------------------
struct Ctx {
    // The problem header (state, goal, weights, theta_0 phasor, padding multiplicities) and the table of f64
    // literals are kept in LDS and read where they are used: as SGPR residents they overflowed the scalar file and
    // every spilled access became a v_readlane on the (saturated) VALU.
    const double* hd;   // [64]: header slots H_*, then KTAB at KC_BASE
    double* akkt;       // one double of LDS owned by this wavefront: the inner tolerance of the running inner problem (HD_AKKT)
    bool pad_f, pad_d, terminal;  // any zero-padded other-robot / dynamic rows; non-zero terminal weights
    int Ks, Kf, Kd;
    // lane roles
    int lane, ik, isub;
    bool vl, il;
    double vref;  // vector lane k: speed reference of step k
    // LDS tables
    double *seg, *stc, *fxy, *dyn, *dync, *pos, *H, *W, *part, *stash;
    double* bal;               // balanced walk: (gx, gy) accumulators of the single-lane steps
    int balT, balT2;           // ... its trip counts (from Kd): every lane makes balT trips, the two-lane steps balT2 of their own
    const double* dynl;        // linear centre tables: [Kd][DYNL]
    const unsigned* dynr;      // ... and the residuals of every (row, step): signed 16-bit multiples of the row's unit, x low, y high
};
// balanced walk of the dynamic rows (eval_point): the two-lane steps own ceil(Kd / 2) trips; T = the smallest trip count at which
// the rows T.. of the single-lane steps fit into the spare trips of the helper lanes.  (Every kernel that runs eval_point: load_problem,
// and the latency kernel's own context set-up.)
__device__ __forceinline__ void set_balanced_trips(Ctx& cx, int N, int lanes) {
    const int ns1 = N - (lanes - N), nh = 2 * (lanes - N);
    int t2 = (cx.Kd + 1) / 2, T = t2;
    while (ns1 > 0 && (cx.Kd - T) * ns1 > nh * (T - t2)) ++T;
    cx.balT = T; cx.balT2 = t2;
}
constexpr int KC_BASE = 32;
constexpr int HD_AKKT = 59;   // LDS header (slots 59..63 are free behind the literal table; the latency kernel uses a slot per wavefront): the inner tolerance eps_nu of the running
                              // inner problem.  It changes ten times per solve and is read once per PANOC step, next to a square root on
                              // the decision path: as a solver variable the 128-VGPR build kept it in a VGPR, spilled it and reloaded
                              // it from scratch in every step (512 B per step and wavefront through an L2 the L-BFGS rings overflow).
#define KC(i) (cx.hd[KC_BASE + (i)])
#define HD(i) (cx.hd[(i)])

struct EvalOut {
    double psi;             // uniform
    double f, nrm2F2;       // uniform; valid only when the evaluation was asked for them (want_f)
    double gv, gw;          // vector lanes: d psi / d (v_k, w_k)
    double F1a, F1b;        // vector lanes: acceleration mapping F1[k], F1[N+k]
    double F2e, F2pad;      // lane i < Kd: F2 of dynamic entry i; uniform: F2 of every padded row
};

template <int NT, bool SC, class P, bool LIN = false, int MINW = 3>
__device__ __forceinline__ void load_problem(const KParams& kp, const double* __restrict__ ws, double* lds,
                                             Ctx& cx) {
    static_assert(!LIN || SC, "linear centre tables belong to the shape-constant form");
    const int lane = P::lane();  // lane inside the problem; `lds` is this problem's carve
    const int N = NT ? NT : kp.N;
    cx.lane = lane;
    cx.vl = lane < N;
    cx.il = true;
    cx.ik = lane % N;
    cx.isub = lane / N;
    // compiled horizon, one problem per wavefront: the fixed part of the carve sits at compile-time offsets (fixed_lds)
    constexpr bool FIXED = NT != 0 && !P::DUO;
    constexpr FixedLds FL = fixed_lds(NT ? NT : 2, MemOf<NT>::value ? MemOf<NT>::value : 1, false, MINW);  // fields up to `gg` do not depend on where S, Y live
    double* hd = lds + (FIXED ? FL.hd : kp.l_hd);
    if (lane < KC_BASE) hd[lane] = ws[lane];
    if (lane < 27) hd[KC_BASE + lane] = KTAB[lane];
    cx.hd = hd;
    cx.akkt = hd + HD_AKKT;
    auto U = [&](int i) { return P::uni(ws[i]); };
    cx.Ks = (int)U(H_KS); cx.Kf = (int)U(H_KF); cx.Kd = (int)U(H_KD);
    cx.pad_f = U(H_NPF) > 0.0; cx.pad_d = U(H_NPD) > 0.0;
    cx.terminal = U(H_QN) != 0.0 || U(H_QTHN) != 0.0;
    cx.vref = cx.vl ? ws[kp.ws_vref + lane] : 0.0;
    cx.stc = lds + kp.l_stc; cx.fxy = lds + kp.l_fxy;
    cx.dyn = lds + kp.l_dyn; cx.dync = lds + kp.l_dync;
    cx.dynl = lds + kp.l_dynl; cx.dynr = reinterpret_cast<const unsigned*>(lds + kp.l_dyn);   // LIN: the residuals take the place of the centres
    if (FIXED) {
        cx.seg = lds + FL.seg; cx.pos = lds + FL.pos; cx.stash = lds + FL.stash; cx.H = lds + FL.part; cx.W = lds + FL.W; cx.part = lds + FL.part;
        cx.bal = lds + FL.bal;
    } else {
        cx.seg = lds + kp.l_seg; cx.pos = lds + kp.l_pos; cx.stash = lds + kp.l_stash; cx.H = lds + kp.l_H; cx.W = lds + kp.l_W; cx.part = lds + kp.l_part;
        cx.bal = lds + kp.l_bal;
    }
    set_balanced_trips(cx, N, P::W);
    // coalesced table copies HBM -> LDS (only the active entries of this problem)
    for (int i = lane; i < N * SEGW; i += P::W) cx.seg[i] = ws[kp.ws_seg + i];
    for (int i = lane; i < cx.Ks * STCW; i += P::W) cx.stc[i] = ws[kp.ws_stc + i];
    for (int i = lane; i < cx.Kf * N * 2; i += P::W) cx.fxy[i] = ws[kp.ws_fxy + i];
    if (SC) {
        // shape-constant rows: per-row constants from the step-0 record (+ alpha), per-step centre, per-step q_dyn
        for (int i = lane; i < cx.Kd * DYNC; i += P::W) {
            const int r = i / DYNC, f = i - r * DYNC;
            cx.dync[i] = f < 6 ? ws[kp.ws_dyn + (r * N) * DYNW + 2 + f] : ws[kp.ws_alpha + r];
        }
        if (LIN) {
            for (int i = lane; i < cx.Kd * DYNL; i += P::W) (lds + kp.l_dynl)[i] = ws[kp.ws_dynl + i];
            for (int i = lane; i < cx.Kd * N; i += P::W)
                reinterpret_cast<unsigned*>(lds + kp.l_dyn)[i] = reinterpret_cast<const unsigned*>(ws + kp.ws_dynr)[i];
        } else {
            for (int i = lane; i < cx.Kd * N; i += P::W) {
                const double* d = ws + kp.ws_dyn + i * DYNW;
                cx.dyn[i * DYNP] = d[0]; cx.dyn[i * DYNP + 1] = d[1];
            }
        }
        // q_dyn of step i: the pad double of segment record i (after the copy of the records above: LDS writes keep their order)
        for (int i = lane; i < N; i += P::W) cx.seg[SEGW * i + 8] = ws[kp.ws_qd + i];
    } else {
        for (int i = lane; i < cx.Kd * N * DYNW; i += P::W) cx.dyn[i] = ws[kp.ws_dyn + i];
    }
    wave_sync();
}

// one dynamic-obstacle item: ellipse frame coordinates and inverse squared semi-axes
struct DynItem {
    double a, b, ca, sa, ihx, ihy, isx, isy, wgt;
};
// AXIS: every active dynamic row of the batch is an axis-aligned ellipse (angle 0, i.e. cos = 1 and sin = 0 exactly -- what the
// reference's own prediction feeder produces, src/main.py:77-85): the rotation into the ellipse frame is the identity up to the
// sign of b.  The general expressions give a = ex and b = -ey EXACTLY in that case (x*1 + y*0), so skipping them changes no bit.
template <bool SC, bool AXIS = false, bool LIN = false>
__device__ __forceinline__ DynItem dyn_item(const Ctx& cx, int i, int k, int N, double px, double py) {
    DynItem d;
    double ex, ey;
    if (SC) {
        const double* s = cx.dync + i * DYNC;
        if (LIN) {   // centre = fma(n, unit, predictor): bitwise the stored centre (prep_problem)
            const double* L = cx.dynl + i * DYNL;
            const int r = (int)cx.dynr[i * N + k];
            const double kk = (double)k;
            const double ecx = __builtin_fma((double)((r << 16) >> 16), L[4], __builtin_fma(L[2], kk, L[0]));
            const double ecy = __builtin_fma((double)(r >> 16), L[5], __builtin_fma(L[3], kk, L[1]));
            ex = px - ecx; ey = py - ecy;
        } else {
            const double* e = cx.dyn + (i * N + k) * DYNP;
            ex = px - e[0]; ey = py - e[1];
        }
        d.wgt = cx.seg[SEGW * k + 8] * s[6];  // q_dyn[k] * alpha: the same product prep_kernel forms for the general table
        d.ca = s[0]; d.sa = s[1]; d.ihx = s[2]; d.ihy = s[3]; d.isx = s[4]; d.isy = s[5];
    } else {
        const double* e = cx.dyn + (i * N + k) * DYNW;
        ex = px - e[0]; ey = py - e[1];
        d.ca = e[2]; d.sa = e[3]; d.ihx = e[4]; d.ihy = e[5]; d.isx = e[6]; d.isy = e[7]; d.wgt = e[8];
    }
    if (AXIS) { d.a = ex; d.b = -ey; }
    else {
        d.a = ex * d.ca + ey * d.sa;
        d.b = ex * d.sa - ey * d.ca;
    }
    return d;
}

// ------------------------------------------------------------------------------------------------
// psi(u; c, y), f(u), F1, F2 and (optionally) grad psi at the point held by the vector lanes.
// ------------------------------------------------------------------------------------------------
template <int NT, bool SC, class P, bool AXIS = false, bool LIN = false, int MINW = 3>
__device__ __forceinline__ void eval_point(const KParams& kp, const Ctx& cx, double v, double w, double c, double icm,
                                           double ya, double yb, bool want_grad, bool want_f, EvalOut& out PROF_ARG) {
    const int N = NT ? NT : kp.N;
    int lane = cx.lane;
    asm volatile("" : "+v"(lane));  // opaque: per-lane LDS addresses are rebuilt here instead of being hoisted out
                                    // of the solver loop, where they would stay live across the whole iteration
    // item lanes: lane = sub*N + k.  Step k is served by LPS = (63-k)/N + 1 lanes (3-4 at N = 20; at N = 40 the 24 spare
    // lanes double up on the first 24 steps, which also carry the most reference segments)
    // When 64 is almost a multiple of N (N = 20: 60 lanes) a uniform split is cheaper: the loop strides are constants.
    constexpr int PW = P::W;  // lanes of this problem
    constexpr bool UNIFORM = NT != 0 && (PW % NT) * 5 <= NT;
    const bool c_vl = lane < N;
    const bool c_il = UNIFORM ? lane < (PW / N) * N : true;
    const int c_ik = lane % N, c_isub = lane / N;
    const int LPS = UNIFORM ? PW / N : (PW - 1 - c_ik) / N + 1;
    constexpr int RV = P::RV, RI = P::RI;
    const int STW = NT ? stash_stride_c(NT, MemOf<NT>::value, MINW) : stash_stride_c(kp.N, kp.mem);   // stash doubles per step
    const double ts = here_s(kp.ts);
    const double fleetw = here_s(kp.fleetw);
    const double inf = __builtin_huge_val();
    if (!c_vl) { v = 0.0; w = 0.0; }
    if (NT == 20) MPC_PRIO_CHAIN();

    // ---- rollout.  Heading phasors e^{i theta}: theta_{k+1} = theta_k + ts*w_k, so they are a prefix PRODUCT of
    //      unit complex numbers e^{i ts w_k} (DPP scan); positions are a prefix SUM of Simpson increments.
    double c0, s0, cm, sm, c2, s2;
    {
        const double hd = 0.5 * ts * w;  // half-step heading increment
        // The polynomial path runs unconditionally; the test whether it was admissible (it is, unless a trial point lies far
        // outside the input box) is evaluated next to it instead of in front of it, so that no LDS read + ballot + branch sits
        // at the head of every evaluation's dependency chain.
        {
            double sh, ch;
            sincos_small(hd, cx.hd + KC_BASE, sh, ch);
            double er = ch * ch - sh * sh, ei = 2.0 * sh * ch;  // e^{i ts w_k}
            P::template cprod<RV>(er, ei);                          // prod_{j<=k} e^{i ts w_j}
            c2 = HD(H_CTH0) * er - HD(H_STH0) * ei;                      // heading k+1
            s2 = HD(H_CTH0) * ei + HD(H_STH0) * er;
            c0 = shift_up1(c2, lane, HD(H_CTH0));
            s0 = shift_up1(s2, lane, HD(H_STH0));
            cm = c0 * ch - s0 * sh;
            sm = c0 * sh + s0 * ch;
        }
        if (P::any(fabs(hd) > KC(K_SMALL))) {  // rare: plain sincos of the summed angles
            const double tw = ts * w;
            const double th1 = HD(H_TH0) + P::template prefix<RV>(tw);
            sincos(th1 - 0.5 * tw, &sm, &cm);
            sincos(th1, &s2, &c2);
            c0 = shift_up1(c2, lane, HD(H_CTH0));
            s0 = shift_up1(s2, lane, HD(H_STH0));
        }
    }
#if MPC_DUP_SHARED
    // MEASUREMENT BUILD (round 5, `make variants`: libmpcgpu_dupshared.so): what could "two evaluation points per pass" save at
    // best?  A pass that carries the half step of the Lipschitz test next to the first line-search trial (rows 2-3) would share the
    // heading chain -- half-angle polynomials, the complex prefix product, the rotations -- between the two points; everything
    // behind it (positions, item phases, per-step terms, psi) runs once per point.  Here the gradient-free evaluation (the
    // Lipschitz test: one per PANOC step) executes that chain TWICE, on an opaque copy of its input, and throws the copy away: the
    // time this build loses is the time the pairing could win.  Same results (profiles/r05_two_point_ceiling.txt).
    if (!want_grad) {
        double w2 = w;
        asm volatile("" : "+v"(w2));
        const double hd2 = 0.5 * ts * w2;
        double sh, ch;
        sincos_small(hd2, cx.hd + KC_BASE, sh, ch);
        double er = ch * ch - sh * sh, ei = 2.0 * sh * ch;
        P::template cprod<RV>(er, ei);
        const double d2 = HD(H_CTH0) * er - HD(H_STH0) * ei, e2 = HD(H_CTH0) * ei + HD(H_STH0) * er;
        const double d0 = shift_up1(d2, lane, HD(H_CTH0)), e0 = shift_up1(e2, lane, HD(H_STH0));
        const double dm = d0 * ch - e0 * sh, em = d0 * sh + e0 * ch;
        asm volatile("" :: "v"(d2), "v"(e2), "v"(dm), "v"(em));
    }
#endif
    PROF_MARK(0);  // headings
    const double sixth = KC(K_SIXTH);
    double kCx = 0.0, kSy = 0.0, kdCw = 0.0, kdSw = 0.0;   // stash stride 0: the Simpson values stay here
    {
    const double Cx = (c0 + 4.0 * cm + c2) * sixth, Sy = (s0 + 4.0 * sm + s2) * sixth;
    const double dCw = -ts * (2.0 * sm + s2) * sixth, dSw = ts * (2.0 * cm + c2) * sixth;
    if (STW == 0) { kCx = Cx; kSy = Sy; kdCw = dCw; kdSw = dSw; }
    double pX, pY;
    // lanes beyond the horizon carry v = 0 and finite phasors: their increments are (signed) zeros without a select, and an
    // inclusive PREFIX never reads them into a vector lane
    P::template prefix2<RV>(ts * v * Cx, ts * v * Sy, pX, pY);
    const double X = HD(H_X0) + pX;
    const double Y = HD(H_Y0) + pY;
    if (lane < cx.Kd) cx.H[lane] = zero_here();   // row sums of the hard-constraint hinges, accumulated by the item lanes below
    if (bal_doubles_c(NT ? NT : 2, MemOf<NT>::value ? MemOf<NT>::value : 1) && !P::DUO && want_grad &&
        lane < bal_doubles_c(NT ? NT : 2, MemOf<NT>::value ? MemOf<NT>::value : 1))
        cx.bal[lane] = zero_here();                // balanced walk: gradient accumulators of the single-lane steps
    if (c_vl) {
        cx.pos[2 * lane] = X; cx.pos[2 * lane + 1] = Y;
        // rollout quantities needed again only after the item phase (per-step terms, adjoint): parked in LDS so
        // that they do not occupy registers across the item loops
        if (STW >= 4) {
            double* st = cx.stash + lane * STW;
            st[0] = Cx; st[1] = Sy; st[2] = dCw; st[3] = dSw;
            if (STW == 6) { st[4] = v; st[5] = w; }
        }
    }
    }
    wave_sync();
    PROF_MARK(1);  // positions + publish
    if (NT == 20) MPC_PRIO_ITEMS();

    // ---- item phase A: stage terms of step ik handled by this lane
    double cost_l = 0.0, S_l = 0.0, gx = 0.0, gy = 0.0, dsx = 0.0, dsy = 0.0;
    double best = inf, bgx = 0.0, bgy = 0.0;
    double px = 0.0, py = 0.0;
    bool anyh = false;
    // Dynamic rows are visited in TRIPS (trip t: row c_isub + t LPS of every item lane).  Bit t of hmask: some item lane of trip t
    // lies inside a hard ellipse -- the second item pass (weighted hard-constraint gradients) then skips the trips in which no
    // lane has anything to add: the obstacle that is being crossed sits in one or two rows, the others are far away.  Kept where
    // the steps of the second half of the horizon have ONE item lane each and a trip is a single row (N_hor = 40: -1.5 % kernel
    // time); with three lanes per step (N_hor = 20: three trips of three rows) the ballots cost what the skipped trips save
    // (+1 %, measured) and the rows are walked as in rounds 1-3.
    constexpr bool HM = !UNIFORM;
    unsigned hmask = 0u;
    // MPC_CACHE_B: d(Ih)/d(position) of this lane's first three rows, zero outside the hard ellipse (see the knob's comment)
    constexpr bool CB = MPC_CACHE_B != 0 && UNIFORM && !P::DUO && NT != 0;
    constexpr bool CB40 = MPC_CACHE_B40 != 0 && NT != 0 && !UNIFORM && !P::DUO && balanced_shape(NT ? NT : 2, MemOf<NT>::value ? MemOf<NT>::value : 1);
    double cbx0 = 0.0, cby0 = 0.0, cbx1 = 0.0, cby1 = 0.0, cbx2 = 0.0, cby2 = 0.0, cbx3 = 0.0, cby3 = 0.0;
    const int minLPS = UNIFORM ? PW / N : (PW - N) / N + 1;
    // BALANCED WALK (round 4, N_hor = 40).  Steps 0 .. NL2-1 have two item lanes, the NS1 steps behind them ONE: walking the rows
    // step by step, those lanes make Kd trips while the others are done after Kd / 2 -- and the wavefront waits for them.  Here
    // every lane makes balT trips: the lane of a single-lane step takes rows 0 .. balT-1 of its step, the NH lanes of the two-lane
    // steps take their own ceil(Kd / 2) rows and then FOREIGN items -- rows balT.. of the single-lane steps (Kd = 8: five trips
    // instead of eight).  A foreign item reads the position of its step from LDS and adds its gradient to that step's
    // accumulator in LDS (cx.bal; ds_add_f64, rare: only inside an ellipse); cost terms are wave-summed anyway and the hinge
    // row sums are LDS atomics already.  Another summation order than the row walk for the steps of the second part of the horizon.
    constexpr bool BAL = NT != 0 && !UNIFORM && !P::DUO && balanced_shape(NT ? NT : 2, MemOf<NT>::value ? MemOf<NT>::value : 1);
    constexpr int NL2 = NT ? PW - NT : 0, NS1 = NT ? NT - NL2 : 1, NH = 2 * NL2;
    const bool owner1 = BAL && lane >= NL2 && lane < N;      // the only item lane of its step
    const int hid = lane < NL2 ? lane : lane - N + NL2;      // helper index 0 .. NH-1 of the lanes of the two-lane steps
    const int ntrip = BAL ? cx.balT : HM ? (cx.Kd + minLPS - 1) / minLPS : 0;
    // (row, step, position, foreign?) of this lane's item in trip t
    auto trip_item = [&](int t, int k_own, double px_own, double py_own, int& i, int& kk, double& qx, double& qy) -> bool {
        i = c_isub + t * LPS; kk = k_own; qx = px_own; qy = py_own;      // (a single-lane step: LPS = 1, row t)
        if (BAL && !owner1) {                                            // called for the trips t >= balT2 only
            const int f = hid + NH * (t - cx.balT2);
            i = cx.balT + f / NS1; kk = NL2 + f % NS1;
            if (i < cx.Kd) { qx = cx.pos[2 * kk]; qy = cx.pos[2 * kk + 1]; }
            return true;
        }
        return false;
    };
    if (c_il) {
        const int k = c_ik;
        px = cx.pos[2 * k]; py = cx.pos[2 * k + 1];
        // reference-path deviation: min over segments i >= k (mpc_generator.py:207,116-130,28-36).
        // Exact pruning: segments inside a circle that is farther from the robot than sqrt(best) cannot hold the minimum
        // (nor tie with it), so their distances are not evaluated; the value and the arg-min are unchanged.
        double sb = inf;  // upper bound of sqrt(best)
        // (1) the SEG_WIN * LPS segments of this step nearest in index are always evaluated
        int i = k + c_isub;
        MPC_ITEM_LOOP
        for (int it = 0; it < SEG_WIN && i < N; ++it, i += LPS) {
            const double* sg = cx.seg + SEGW * i;
            const double s1x = sg[0], s1y = sg[1], dx = sg[2], dy = sg[3], inv = sg[4];
            const double th = ((px - s1x) * dx + (py - s1y) * dy) * inv;
            const double t = clampd(th, 0.0, 1.0);
            const double wx = s1x + t * dx - px, wy = s1y + t * dy - py;
            const double d2 = wx * wx + wy * wy;
            if (d2 < best) {
                best = d2;
                const double wd = (th >= 0.0 && th <= 1.0) ? (wx * dx + wy * dy) * inv : 0.0;
                bgx = 2.0 * (wd * dx - wx);
                bgy = 2.0 * (wd * dy - wy);
            }
        }
        // (2) every later segment of this lane lies in the bounding circle stored with segment i (the circle of ALL segments
        //     from i to the end of the path): while that circle is within sqrt(best) of the robot, segment i is evaluated; as
        //     soon as it is not, nothing from i on can hold the minimum (nor tie with it) and the lane is done.  The suffix
        //     circles recede along the path, so a lane leaves after the few segments that are actually near its position.
        bool more = false;
        if (i < N) {
            sb = sqrt_upper(best, KC(K_SQRT));
            const double* sg = cx.seg + SEGW * i;
            const double bx = px - sg[5], by = py - sg[6], reach = (sb + sg[7]) * KC(K_REACH);
            more = bx * bx + by * by < reach * reach;
        }
        if (P::any(more)) {
            PROF_COUNT(23);
            MPC_ITEM_LOOP
            while (more) {
                const double* sg = cx.seg + SEGW * i;
                const double s1x = sg[0], s1y = sg[1], dx = sg[2], dy = sg[3], inv = sg[4];
                const double th = ((px - s1x) * dx + (py - s1y) * dy) * inv;
                const double t = clampd(th, 0.0, 1.0);
                const double wx = s1x + t * dx - px, wy = s1y + t * dy - py;
                const double d2 = wx * wx + wy * wy;
                if (d2 < best) {
                    best = d2;
                    sb = sqrt_upper(d2, KC(K_SQRT));
                    const double wd = (th >= 0.0 && th <= 1.0) ? (wx * dx + wy * dy) * inv : 0.0;
                    bgx = 2.0 * (wd * dx - wx);
                    bgy = 2.0 * (wd * dy - wy);
                }
                i += LPS;
                more = false;
                if (i < N) {
                    const double* sn = cx.seg + SEGW * i;
                    const double bx = px - sn[5], by = py - sn[6], reach = (sb + sn[7]) * KC(K_REACH);
                    more = bx * bx + by * by < reach * reach;
                }
            }
        }
        PROF_MARK(2);  // segments
        // fleet discs (mpc_generator.py:211-216,105-108)
        MPC_ITEM_LOOP
        for (int j = c_isub; j < cx.Kf; j += LPS) {
            const double ex = px - cx.fxy[(j * N + k) * 2], ey = py - cx.fxy[(j * N + k) * 2 + 1];
            const double hh = kp.W2 - (ex * ex + ey * ey);
            if (hh > 0.0) {
                cost_l += fleetw * hh;
                gx -= 2.0 * fleetw * ex;
                gy -= 2.0 * fleetw * ey;
            }
        }
        // static polygons, 4 half-planes each (mpc_generator.py:219-225,46-54)
        MPC_ITEM_LOOP
        for (int o = c_isub; o < cx.Ks; o += LPS) {
            const double* s = cx.stc + STCW * o;
            const double h0 = s[0] - s[4] * px - s[8] * py, h1 = s[1] - s[5] * px - s[9] * py;
            const double h2 = s[2] - s[6] * px - s[10] * py, h3 = s[3] - s[7] * px - s[11] * py;
            // a point outside (some half-plane value <= 0) has the product exactly 0: the squares and products are formed only in
            // the trips in which some lane is inside a polygon
            if (!P::any(fmin(fmin(h0, h1), fmin(h2, h3)) > 0.0)) continue;
            const double m0 = fmax(0.0, h0), m1 = fmax(0.0, h1), m2 = fmax(0.0, h2), m3 = fmax(0.0, h3);
            const double q0 = m0 * m0, q1 = m1 * m1, q2 = m2 * m2, q3 = m3 * m3;
            const double p01 = q0 * q1, p23 = q2 * q3;
            const double prod = p01 * p23;
            if (prod > 0.0) {
                S_l += prod;
                const double r0 = 2.0 * m0 * q1 * p23, r1 = 2.0 * m1 * q0 * p23;
                const double r2 = 2.0 * m2 * q3 * p01, r3 = 2.0 * m3 * q2 * p01;
                dsx -= r0 * s[4] + r1 * s[5] + r2 * s[6] + r3 * s[7];
                dsy -= r0 * s[8] + r1 * s[9] + r2 * s[10] + r3 * s[11];
            }
        }
        PROF_MARK(3);  // fleet + static
        // dynamic ellipses: hard indicator -> H, soft cost with social margin (mpc_generator.py:229-241,38-44,85-95)
        // one body for both walks: HM: t = 0 .. ntrip-1 for every lane, row c_isub + t LPS if it exists; else i = c_isub, c_isub + LPS ...
        if (BAL) {
            // one (row, step) item of phase A; `foreign`: the step is not this lane's own -- its gradient goes to the step's accumulator
            auto dyn_a = [&](int i, int kk, double qx, double qy, bool foreign, double& tx, double& ty) -> bool {
                const DynItem d = dyn_item<SC, AXIS, LIN>(cx, i, kk, N, qx, qy);
                const double a2 = d.a * d.a, b2 = d.b * d.b;
                const double Ih = 1.0 - a2 * d.ihx - b2 * d.ihy;
                if (Ih > 0.0) {
                    lds_add(cx.H + i, Ih);
                    if (CB40 && want_grad) {   // the factors phase B multiplies the row weight with (own rows of the first trips)
                        if (AXIS) { tx = -2.0 * d.a * d.ihx; ty = 2.0 * d.b * d.ihy; }
                        else { tx = -2.0 * d.a * d.ca * d.ihx - 2.0 * d.b * d.sa * d.ihy; ty = -2.0 * d.a * d.sa * d.ihx + 2.0 * d.b * d.ca * d.ihy; }
                    }
                }
                const double Is = 1.0 - a2 * d.isx - b2 * d.isy;
                if (Is > 0.0) {
                    cost_l += d.wgt * Is * Is;
                    const double wI = 2.0 * d.wgt * Is;
                    if (!foreign) {   // the expressions of the row walk, term for term
                        if (AXIS) {
                            gx += wI * (-2.0 * d.a * d.isx);
                            gy += wI * (2.0 * d.b * d.isy);
                        } else {
                            gx += wI * (-2.0 * d.a * d.ca * d.isx - 2.0 * d.b * d.sa * d.isy);
                            gy += wI * (-2.0 * d.a * d.sa * d.isx + 2.0 * d.b * d.ca * d.isy);
                        }
                    } else if (want_grad) {
                        double* acc = cx.bal + 2 * (kk - NL2);
                        if (AXIS) {
                            lds_add(acc, wI * (-2.0 * d.a * d.isx));
                            lds_add(acc + 1, wI * (2.0 * d.b * d.isy));
                        } else {
                            lds_add(acc, wI * (-2.0 * d.a * d.ca * d.isx - 2.0 * d.b * d.sa * d.isy));
                            lds_add(acc + 1, wI * (-2.0 * d.a * d.sa * d.isx + 2.0 * d.b * d.ca * d.isy));
                        }
                    }
                }
                return Ih > 0.0;
            };
            // trips 0 .. balT2-1: every lane walks rows of its own step (the row walk, unchanged) ...
            double ux, uy;   // factors nobody keeps
            auto own_trip = [&](int t, int i, double& tx, double& ty) {
                bool inside = false;
                if (i < cx.Kd) inside = dyn_a(i, k, px, py, false, tx, ty);
                anyh |= inside;
                if (P::any(inside)) hmask |= 1u << t;
            };
            int t_own = 0, i_own = c_isub;
            if (CB40) {
                if (t_own < cx.balT2) { own_trip(t_own, i_own, cbx0, cby0); ++t_own; i_own += LPS; }
                if (MPC_CACHE_B40 >= 2 && t_own < cx.balT2) { own_trip(t_own, i_own, cbx1, cby1); ++t_own; i_own += LPS; }
                if (MPC_CACHE_B40 >= 3 && t_own < cx.balT2) { own_trip(t_own, i_own, cbx2, cby2); ++t_own; i_own += LPS; }
                if (MPC_CACHE_B40 >= 4 && t_own < cx.balT2) { own_trip(t_own, i_own, cbx3, cby3); ++t_own; i_own += LPS; }
            }
            MPC_ITEM_LOOP
            for (; t_own < cx.balT2; ++t_own, i_own += LPS) own_trip(t_own, i_own, ux, uy);
            // ... trips balT2 .. balT-1: the single-lane steps go on with their rows, the other lanes take foreign items
            MPC_ITEM_LOOP
            for (int t = cx.balT2; t < ntrip; ++t) {
                bool inside = false;
                int i, kk; double qx, qy;
                const bool foreign = trip_item(t, k, px, py, i, kk, qx, qy);
                if (i < cx.Kd) inside = dyn_a(i, kk, qx, qy, foreign, ux, uy);
                anyh |= inside;
                if (P::any(inside)) hmask |= 1u << t;
            }
        } else {
        // one (row, step) item of phase A; (tx, ty) receive d(Ih)/d(px, py) when the position is inside the hard ellipse (CB)
        auto dyn_a1 = [&](int i, double& tx, double& ty) -> bool {
            const DynItem d = dyn_item<SC, AXIS, LIN>(cx, i, k, N, px, py);
            const double a2 = d.a * d.a, b2 = d.b * d.b;
            const double Ih = 1.0 - a2 * d.ihx - b2 * d.ihy;
            if (Ih > 0.0) {
                lds_add(cx.H + i, Ih);      // D_i = sum_k max(0, Ih(i, k)): accumulated where the terms arise
                if (CB && want_grad) {      // the factors phase B multiplies the row weight W_i with
                    if (AXIS) {
                        tx = -2.0 * d.a * d.ihx;
                        ty = 2.0 * d.b * d.ihy;
                    } else {
                        tx = -2.0 * d.a * d.ca * d.ihx - 2.0 * d.b * d.sa * d.ihy;
                        ty = -2.0 * d.a * d.sa * d.ihx + 2.0 * d.b * d.ca * d.ihy;
                    }
                }
            }
            const double Is = 1.0 - a2 * d.isx - b2 * d.isy;
            if (Is > 0.0) {
                cost_l += d.wgt * Is * Is;
                const double wI = 2.0 * d.wgt * Is;
                if (AXIS) {   // the general terms with cos = 1, sin = 0: the products by 1 are exact, those by 0 vanish
                    gx += wI * (-2.0 * d.a * d.isx);
                    gy += wI * (2.0 * d.b * d.isy);
                } else {
                    gx += wI * (-2.0 * d.a * d.ca * d.isx - 2.0 * d.b * d.sa * d.isy);
                    gy += wI * (-2.0 * d.a * d.sa * d.isx + 2.0 * d.b * d.ca * d.isy);
                }
            }
            return Ih > 0.0;
        };
        if (CB) {
            int i = c_isub;
            if (i < cx.Kd) anyh |= dyn_a1(i, cbx0, cby0);
            i += LPS;
            if (i < cx.Kd) anyh |= dyn_a1(i, cbx1, cby1);
            i += LPS;
            if (i < cx.Kd) anyh |= dyn_a1(i, cbx2, cby2);
            i += LPS;
            double ux, uy;      // rows beyond the cached trips: phase B re-derives them
            MPC_ITEM_LOOP
            for (; i < cx.Kd; i += LPS) anyh |= dyn_a1(i, ux, uy);
        } else {
        MPC_ITEM_LOOP
        for (int t = 0, i = c_isub; HM ? t < ntrip : i < cx.Kd; ++t, i += LPS) {
            bool inside = false;
            if (!HM || i < cx.Kd) {
                double ux, uy;
                inside = dyn_a1(i, ux, uy);
                anyh |= inside;
            }
            if (HM && P::any(inside)) hmask |= 1u << t;
        }
        }
        }
    }
    if (HM) hmask = P::uni_u(hmask);
    PROF_MARK(4);  // dynamic
    // position of this vector lane's step: lane k < N is item lane (k, sub 0), px / py hold it already
    const double X = c_vl ? px : 0.0, Y = c_vl ? py : 0.0;
    // zero-padded rows, closed form on the vector lanes: npf discs of radius W and npd degenerate ellipses
    // (semi-axes 1e-6, alpha = 0: hard indicator only) at the origin
    double hp = 0.0, r2o = 0.0;
    if (c_vl) {
        r2o = X * X + Y * Y;
        if (cx.pad_f) {
            const double hh = kp.W2 - r2o;
            if (hh > 0.0) cost_l += fleetw * HD(H_NPF) * hh;  // its gradient is added on the vector lanes below
        }
        if (cx.pad_d) {
            const double ipad = KC(K_IPAD);
            hp = fmax(0.0, 1.0 - X * X * ipad - Y * Y * ipad);
        }
    }
    const bool any_hp = P::any(hp > 0.0);
    const bool any_h = HM ? hmask != 0u : P::any(anyh);
    wave_sync();

    // ---- constraint sums: S (static, broadcast into every F2 component), D_i (row sums of H).
    //      Every reduction is skipped when the ballots show that all its terms are zero.
    const bool any_S = P::any(S_l > 0.0);
    const double S = any_S ? P::template sum<RI>(S_l) : 0.0;
    double F2e = 0.0;
    if (lane < cx.Kd) {
        double D = 0.0;
        if (any_h) D = cx.H[lane];
        F2e = S + D;
    }
    const double F2pad = cx.pad_d ? S + (any_hp ? P::template sum<RV>(hp) : 0.0) : 0.0;
    const bool viol = any_S || any_h || any_hp;  // some penalty constraint is violated
    out.F2e = F2e;
    out.F2pad = F2pad;

    PROF_MARK(5);  // pads + constraint sums
    // ---- item phase B (only when some penalty constraint is violated): weighted hard-constraint gradients
    double Gpx = 0.0, Gpy = 0.0;  // vector lanes: gradient of the padded-row terms w.r.t. the position
    if (want_grad && viol) {
        // sum of ALL F2 entries: the weight of the static-polygon term S inside every entry.  Its gradient (dsx, dsy) is exactly
        // zero in every lane unless some lane is inside a polygon (any_S): the reduction is formed only then
        const double sumF2 = any_S ? P::template sum<2>(F2e) + HD(H_NPD) * F2pad : 0.0;
        if (lane < cx.Kd) cx.W[lane] = c * F2e;
        wave_sync();
        if (c_il) {
            const int k = c_ik;
            if (any_h && BAL) {
                auto dyn_b = [&](int i, int kk, double qx, double qy, bool foreign) {
                    const DynItem d = dyn_item<SC, AXIS, LIN>(cx, i, kk, N, qx, qy);
                    const double Ih = 1.0 - d.a * d.a * d.ihx - d.b * d.b * d.ihy;
                    if (Ih > 0.0) {
                        const double wi = cx.W[i];
                        if (!foreign) {
                            if (AXIS) {
                                gx += wi * (-2.0 * d.a * d.ihx);
                                gy += wi * (2.0 * d.b * d.ihy);
                            } else {
                                gx += wi * (-2.0 * d.a * d.ca * d.ihx - 2.0 * d.b * d.sa * d.ihy);
                                gy += wi * (-2.0 * d.a * d.sa * d.ihx + 2.0 * d.b * d.ca * d.ihy);
                            }
                        } else {
                            double* acc = cx.bal + 2 * (kk - NL2);
                            if (AXIS) {
                                lds_add(acc, wi * (-2.0 * d.a * d.ihx));
                                lds_add(acc + 1, wi * (2.0 * d.b * d.ihy));
                            } else {
                                lds_add(acc, wi * (-2.0 * d.a * d.ca * d.ihx - 2.0 * d.b * d.sa * d.ihy));
                                lds_add(acc + 1, wi * (-2.0 * d.a * d.sa * d.ihx + 2.0 * d.b * d.ca * d.ihy));
                            }
                        }
                    }
                };
                int t_own = 0, i_own = c_isub;
                if (CB40) {   // the cached trips: W_i times the factor phase A left (zero outside the ellipse)
                    auto cached = [&](double tx, double ty) {
                        if (t_own < cx.balT2) {
                            if (((hmask >> t_own) & 1u) && i_own < cx.Kd) { const double wi = cx.W[i_own]; gx += wi * tx; gy += wi * ty; }
                            ++t_own; i_own += LPS;
                        }
                    };
                    cached(cbx0, cby0);
                    if (MPC_CACHE_B40 >= 2) cached(cbx1, cby1);
                    if (MPC_CACHE_B40 >= 3) cached(cbx2, cby2);
                    if (MPC_CACHE_B40 >= 4) cached(cbx3, cby3);
                }
                MPC_ITEM_LOOP
                for (; t_own < cx.balT2; ++t_own, i_own += LPS) {
                    if (!((hmask >> t_own) & 1u) || i_own >= cx.Kd) continue;          // no lane of this trip is inside a hard ellipse
                    dyn_b(i_own, k, px, py, false);
                }
                MPC_ITEM_LOOP
                for (int t = cx.balT2; t < ntrip; ++t) {
                    if (!((hmask >> t) & 1u)) continue;
                    int i, kk; double qx, qy;
                    const bool foreign = trip_item(t, k, px, py, i, kk, qx, qy);
                    if (i < cx.Kd) dyn_b(i, kk, qx, qy, foreign);
                }
            } else if (any_h && CB) {
                // the cached trips: W_i times the factor phase A left (zero where the position is outside the ellipse: gx + W_i * 0 = gx)
                int i = c_isub;
                if (i < cx.Kd) { const double wi = cx.W[i]; gx += wi * cbx0; gy += wi * cby0; }
                i += LPS;
                if (i < cx.Kd) { const double wi = cx.W[i]; gx += wi * cbx1; gy += wi * cby1; }
                i += LPS;
                if (i < cx.Kd) { const double wi = cx.W[i]; gx += wi * cbx2; gy += wi * cby2; }
                i += LPS;
                MPC_ITEM_LOOP
                for (; i < cx.Kd; i += LPS) {
                    const DynItem d = dyn_item<SC, AXIS, LIN>(cx, i, k, N, px, py);
                    const double Ih = 1.0 - d.a * d.a * d.ihx - d.b * d.b * d.ihy;
                    if (Ih > 0.0) {
                        const double wi = cx.W[i];
                        if (AXIS) {
                            gx += wi * (-2.0 * d.a * d.ihx);
                            gy += wi * (2.0 * d.b * d.ihy);
                        } else {
                            gx += wi * (-2.0 * d.a * d.ca * d.ihx - 2.0 * d.b * d.sa * d.ihy);
                            gy += wi * (-2.0 * d.a * d.sa * d.ihx + 2.0 * d.b * d.ca * d.ihy);
                        }
                    }
                }
            } else if (any_h) {
                MPC_ITEM_LOOP
                for (int t = 0, i = c_isub; HM ? t < ntrip : i < cx.Kd; ++t, i += LPS) {
                    if (HM && (!((hmask >> t) & 1u) || i >= cx.Kd)) continue;   // no lane of this trip is inside a hard ellipse
                    const DynItem d = dyn_item<SC, AXIS, LIN>(cx, i, k, N, px, py);
                    const double Ih = 1.0 - d.a * d.a * d.ihx - d.b * d.b * d.ihy;
                    if (Ih > 0.0) {
                        const double wi = cx.W[i];
                        if (AXIS) {
                            gx += wi * (-2.0 * d.a * d.ihx);
                            gy += wi * (2.0 * d.b * d.ihy);
                        } else {
                            gx += wi * (-2.0 * d.a * d.ca * d.ihx - 2.0 * d.b * d.sa * d.ihy);
                            gy += wi * (-2.0 * d.a * d.sa * d.ihx + 2.0 * d.b * d.ca * d.ihy);
                        }
                    }
                }
            }
            if (any_S) {
                gx += c * sumF2 * dsx;
                gy += c * sumF2 * dsy;
            }
        }
        if (any_hp && hp > 0.0) {  // a = X, b = -Y for the padded ellipse (cosA = 1, sinA = 0)
            const double ipad = KC(K_IPAD);
            const double wpad = c * HD(H_NPD) * F2pad;
            Gpx = wpad * (-2.0 * X * ipad);
            Gpy = wpad * (-2.0 * Y * ipad);
        }
    }

    PROF_MARK(6);  // phase B
    if (NT == 20) MPC_PRIO_CHAIN();
    // ---- combine the LPS item lanes of each step on its vector lane.  Lane k < N is itself the first item lane of step k
    //      (sub 0): its partial stays in registers, only the lanes of sub >= 1 go through LDS.
    if (c_il && c_isub > 0) {
        double* pp = cx.part + ((c_isub - 1) * N + c_ik) * PARTW;
        pp[0] = gx; pp[1] = gy; pp[2] = best; pp[3] = bgx; pp[4] = bgy;
    }
    wave_sync();
    double Gx = Gpx, Gy = Gpy, vcost = 0.0;
    if (c_vl) {
        double bb = inf, wbx = 0.0, wby = 0.0;
        Gx += gx; Gy += gy;
        if (best < bb) { bb = best; wbx = bgx; wby = bgy; }
        for (int s = 1; s < LPS; ++s) {
            const double* pp = cx.part + ((s - 1) * N + lane) * PARTW;
            Gx += pp[0]; Gy += pp[1];
            if (pp[2] < bb) { bb = pp[2]; wbx = pp[3]; wby = pp[4]; }
        }
        if (BAL && want_grad && lane >= NL2) { Gx += cx.bal[2 * (lane - NL2)]; Gy += cx.bal[2 * (lane - NL2) + 1]; }   // foreign items of this step
        Gx += HD(H_QRPD) * wbx; Gy += HD(H_QRPD) * wby;
        vcost = HD(H_QRPD) * bb;
        if (cx.pad_f && kp.W2 - r2o > 0.0) {
            Gx -= 2.0 * fleetw * HD(H_NPF) * X;
            Gy -= 2.0 * fleetw * HD(H_NPF) * Y;
        }
    }

    PROF_MARK(7);  // combine
    // ---- per-step terms on the vector lanes (mpc_generator.py:208-209,246,254-267)
    if (STW == 6) {   // (else (v, w) are still in their registers: zero beyond the horizon since the top of the evaluation)
        v = 0.0; w = 0.0;
        if (c_vl) { v = cx.stash[lane * 6 + 4]; w = cx.stash[lane * 6 + 5]; }
    }
    const double vprev = shift_up1(v, lane, HD(H_VINIT)), wprev = shift_up1(w, lane, HD(H_WINIT));
    // (lane N holds -v_{N-1} / ts here: every consumer of a, bacc -- box distance, costs, adjoint, the outer step -- selects the
    // vector lanes itself)
    const double a = (v - vprev) * kp.inv_ts;
    const double bacc = (w - wprev) * kp.inv_ts;
    out.F1a = a; out.F1b = bacc;
    const double za = a + ya * icm, zb = bacc + yb * icm;
    // z - Proj_C(z): z - amax above the box, z - amin below it, z - z = +0 inside -- the bits of the three-way select, in three
    // instructions per component instead of compares, EXEC masks and branches
    const double ea = za - clamp_u(za, kp.amin, kp.amax);
    const double eb = zb - clamp_u(zb, -kp.aamax, kp.aamax);
    double gthN = 0.0;
    if (c_vl) {
        const double dv = v - cx.vref;
        vcost += HD(H_QVEL) * dv * dv + HD(H_RV) * v * v + HD(H_RW) * w * w + HD(H_ACC) * a * a + HD(H_WACC) * bacc * bacc;
    }
    if (cx.terminal) {  // terminal cost (weights are 0 in the reference's yaml)
        const double thN = HD(H_TH0) + P::template sum<RV>(ts * w);
        if (lane == N - 1) {
            const double ex = X - HD(H_XG), ey = Y - HD(H_YG), et = thN - HD(H_THG);
            vcost += HD(H_QN) * (ex * ex + ey * ey) + HD(H_QTHN) * et * et;
            Gx += 2.0 * HD(H_QN) * ex; Gy += 2.0 * HD(H_QN) * ey;
            gthN = 2.0 * HD(H_QTHN) * et;
        }
        gthN = P::from_lane(gthN, N - 1);
    }
    // psi = f + c/2 dist^2_C(F1 + y/max(c,1)) + c/2 ||F2||^2 in ONE wave reduction of per-lane partials
    // (item lanes: stage costs; vector lanes: per-step costs + box distance; lanes < Kd: their F2 entry)
    const double dist_l = c_vl ? ea * ea + eb * eb : 0.0;
    const double f2_l = viol ? F2e * F2e : 0.0;
    const double pad2 = viol ? HD(H_NPD) * F2pad * F2pad : 0.0;
    out.psi = P::template sum<RI>(cost_l + vcost + 0.5 * c * (dist_l + f2_l)) + 0.5 * c * pad2;
    if (want_f) {  // f and ||F2||^2 on their own (outer loop, test hook): two more reductions
        out.f = P::template sum<RI>(cost_l + vcost);
        out.nrm2F2 = viol ? P::template sum<2>(f2_l) + pad2 : 0.0;
    }

    PROF_MARK(8);  // vector terms + psi
    if (want_grad) {
        const double da = c_vl ? (2.0 * HD(H_ACC) * a + c * ea) * kp.inv_ts : 0.0;
        const double db = c_vl ? (2.0 * HD(H_WACC) * bacc + c * eb) * kp.inv_ts : 0.0;
        const double da_n = shift_down1(da), db_n = shift_down1(db);
        double gv = 2.0 * HD(H_QVEL) * (v - cx.vref) + 2.0 * HD(H_RV) * v + da - da_n;
        double gw = 2.0 * HD(H_RW) * w + db - db_n;
        // adjoint of the rollout: suffix sums instead of a serial backward sweep
        // (kept in registers they are finite beyond the horizon, where v = 0 and the adjoint sums Ax, Ay are 0: same bits)
        double Cx = kCx, Sy = kSy, dCw = kdCw, dSw = kdSw;
        if (STW >= 4 && c_vl) { const double* st = cx.stash + lane * STW; Cx = st[0]; Sy = st[1]; dCw = st[2]; dSw = st[3]; }
        double Ax, Ay;
        P::template suffix2<RV>(Gx, Gy, lane, Ax, Ay);
        const double T = ts * v * (-Sy * Ax + Cx * Ay);   // v = 0 and finite factors beyond the horizon: a (signed) zero there
        const double Bx = P::template suffix<RV>(T, lane) - T;
        gv += ts * (Cx * Ax + Sy * Ay);
        gw += ts * v * (dCw * Ax + dSw * Ay) + ts * (Bx + gthN);
        out.gv = c_vl ? gv : 0.0;
        out.gw = c_vl ? gw : 0.0;
    }
    PROF_MARK(9);  // adjoint
}

// ------------------------------------------------------------------------------------------------
// test-hook kernel: one evaluation per problem through eval_point
// ------------------------------------------------------------------------------------------------
template <int NT, bool SC, class P, bool AXIS = false, bool LIN = false>
__global__ __launch_bounds__(WAVE) void cost_grad_kernel(KParams kp, BatchPtrs io, const double* __restrict__ u,
                                                         const double* __restrict__ xi, double* psi, double* f,
                                                         double* grad, double* F1, double* F2, int B) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int b = P::problem();
    if (b >= B) return;
    const int lane = P::lane(), N = NT ? NT : kp.N;
    const double* ws = io.ws + (size_t)b * kp.ws_stride;
    Ctx cx;
    load_problem<NT, SC, P, LIN>(kp, ws, lds + P::half() * kp.l_total, cx);
    const double* ub = u + (size_t)b * 2 * N;
    const double* xb = xi + (size_t)b * (1 + 2 * N);
    const double v = cx.vl ? ub[2 * lane] : 0.0, w = cx.vl ? ub[2 * lane + 1] : 0.0;
    const double c = xb[0];
    const double ya = cx.vl ? xb[1 + lane] : 0.0, yb = cx.vl ? xb[1 + N + lane] : 0.0;
    EvalOut o;
#ifdef MPC_PROFILE
    Prof prof; prof.start();
#endif
    eval_point<NT, SC, P, AXIS, LIN>(kp, cx, v, w, c, 1.0 / fmax(c, 1.0), ya, yb, true, true, o PROF_PASS);
    if (lane == 0) {
        if (psi) psi[b] = o.psi;
        if (f) f[b] = o.f;
    }
    if (cx.vl) {
        if (grad) { grad[(size_t)b * 2 * N + 2 * lane] = o.gv; grad[(size_t)b * 2 * N + 2 * lane + 1] = o.gw; }
        if (F1) { F1[(size_t)b * 2 * N + lane] = o.F1a; F1[(size_t)b * 2 * N + N + lane] = o.F1b; }
    }
    if (F2 && lane < kp.Ndynobs) {
        const int e = (int)ws[H_ENTRY + lane];
        const double val = __shfl(o.F2e, (threadIdx.x - lane) + (e < 0 ? 0 : e));  // test hook only: one LDS-crossbar gather
        F2[(size_t)b * kp.Ndynobs + lane] = e < 0 ? o.F2pad : val;
    }
}

// ------------------------------------------------------------------------------------------------
// solver kernels: ALM/PM outer loop around PANOC, one problem per wavefront.
// The iteration is organised as a small state machine (initial point, Lipschitz estimate, first step, outer step: the
// evaluations that happen once per inner problem) around the loop of the PANOC steps, which has its own call sites of
// eval_point for the Lipschitz test and the line search (MPC_STEP_LOOP below).
//   solve_kernel_pair: lane k holds (v_k, w_k) of every horizon vector, L-BFGS memory in LDS.
//   (An element-per-lane variant with the L-BFGS memory in 40 VGPRs was built and measured in round 1: the
//   unrolled register-resident two-loop recursion pushed the kernel past 256 VGPRs and it ran 1.6-2.8x slower;
//   see DESIGN.md section 7.)
// ------------------------------------------------------------------------------------------------
enum { ST_INIT0 = 0, ST_INIT1, ST_LIP, ST_NOLS, ST_LS, ST_OUTER };

// 1 (round 3): the PANOC steps of an inner problem run in a loop of their OWN inside the state machine, with one call site of
// eval_point for the Lipschitz test and one for the trial points of the line search (solve_body).  0: rounds 1-3, ONE call site:
// every evaluation is a state of the machine (`make variants` keeps it as libmpcgpu_onesite.so for A/B runs).  Same arithmetic,
// same bits; what changes is what the register allocator has to carry.  With one loop around one call site every variable of the
// solver is a loop-carried value of that loop: on EVERY evaluation ~20 of them were shuffled into their loop-header registers
// (v_mov_b64), four went through scratch (store at the latch, reload at the use -- the ~3 TB of write traffic per launch the
// rocprof passes of rounds 1-3 showed) and 18 SGPR state words were copied.  In the step loop the line search carries (tau, the
// trial point) and nothing else; u, gamma fpr, the direction and the step's scalars are loop-invariant around it.  Measured
// (profiles/archive/r03_step_loop_ab.txt): 1018 -> 896 ms at B = 32 768 (-12 %), 322 -> 281 ms at B = 8192, code 27 -> 44 KB.
#ifndef MPC_STEP_LOOP
#define MPC_STEP_LOOP 1
#endif
#ifndef MPC_MIN_WAVES
#define MPC_MIN_WAVES 3  // waves per SIMD the register allocator must leave room for (512 / MPC_MIN_WAVES VGPRs)
#endif

// ------------------------------------------------------------------------------------------------
// The steps of the iteration [OpEn], shared by every solver kernel (solve_body below, mpc_team.hpp): the same device
// function on the same inputs gives the same bits.  (The library is built with -ffp-contract=on: an expression is
// contracted the way it is written, whatever code surrounds it after inlining.)
// Each use of a double literal reads the LDS table (KC): a local copy would live in two VGPRs across the whole loop.
// ------------------------------------------------------------------------------------------------
template <class P, int ROWS>
__device__ __forceinline__ double dot2r(double a0, double a1, double b0, double b1) {
    return P::template sum<ROWS>(__builtin_fma(a0, b0, a1 * b1));
}
// perturbation of the local Lipschitz estimate: h_i = max(delta, eps * u_i); nh = ||h||
template <class P>
__device__ __forceinline__ void panoc_lip_perturbation(const Ctx& cx, bool vl, double uv, double uw, double& h0, double& h1, double& nh) {
    h0 = vl ? ((KC(K_EPS_LIP) * uv > KC(K_DELTA_LIP)) ? KC(K_EPS_LIP) * uv : KC(K_DELTA_LIP)) : 0.0;
    h1 = vl ? ((KC(K_EPS_LIP) * uw > KC(K_DELTA_LIP)) ? KC(K_EPS_LIP) * uw : KC(K_DELTA_LIP)) : 0.0;
    nh = P::uni(sqrt(dot2r<P, P::RV>(h0, h1, h0, h1)));
}
// L = ||grad(u + h) - grad(u)|| / ||h||, gamma = 0.95 / L, sigma = 0.05 / (4 gamma).
// `ig` = 1 / gamma is carried next to gamma: the step used gamma in the denominator six times (envelope of every trial point and of the iterate,
// sigma, Lipschitz test) and a double-precision division is ~30 VALU instructions on the critical path of a decision; gamma
// itself changes only when the Lipschitz estimate does (halving gamma doubles ig exactly).
template <class P>
__device__ __forceinline__ void panoc_lip_estimate(const Ctx& cx, double d0, double d1, double nh, double& Lip, double& gamma, double& ig,
                                                   double& sigma) {
    Lip = P::uni(sqrt(dot2r<P, P::RV>(d0, d1, d0, d1)) / nh);
    gamma = P::uni(KC(K_GAMMA_L) / fmax(Lip, KC(K_MIN_L)));
    ig = P::uni(1.0 / gamma);
    sigma = P::uni(KC(K_SIGMA) * ig);
}
// u_half <- Proj_U(base - gamma * grad); returns this lane's share of ||gradient_step - u_half||^2
__device__ __forceinline__ double panoc_half_step(const KParams& kp, bool vl, double bv, double bw, double gamma, double g0, double g1,
                                                  double& hv, double& hw) {
    const double sv = bv - gamma * g0, sw = bw - gamma * g1;
    hv = vl ? clamp_u(sv, kp.vmin, kp.vmax) : 0.0;
    hw = vl ? clamp_u(sw, -kp.wmax, kp.wmax) : 0.0;
    const double e0 = vl ? sv - hv : 0.0, e1 = vl ? sw - hw : 0.0;
    return __builtin_fma(e0, e0, e1 * e1);
}
// half step from (bv, bw) plus the two sums of the forward-backward envelope there, ||grad||^2 and
// ||gradient_step - u_half||^2, in ONE packed reduction
template <class P>
__device__ __forceinline__ void panoc_envelope_sums(const KParams& kp, bool vl, double bv, double bw, double gamma, double g0, double g1,
                                                    double& hv, double& hw, double& gg, double& d2h) {
    const double e2 = panoc_half_step(kp, vl, bv, bw, gamma, g0, g1, hv, hw);
    P::sum2(__builtin_fma(g0, g0, g1 * g1), e2, gg, d2h);
}
__device__ __forceinline__ double panoc_fbe(double cost, double gamma, double ig, double gg, double d2h) {
    return cost - 0.5 * gamma * gg + 0.5 * d2h * ig;
}
__device__ __forceinline__ double panoc_fbe_rhs(double cost, double gamma, double ig, double gg, double d2h, double sigma, double nfpr) {
    return panoc_fbe(cost, gamma, ig, gg, d2h) - sigma * nfpr * nfpr;
}
__device__ __forceinline__ double panoc_trial(double u, double r, double d, double tau) { return u - (1.0 - tau) * r - tau * d; }
// Lipschitz test: psi(u_half) > psi(u) + eps |psi(u)| - <grad, gamma fpr> + (0.95 / (2 gamma)) ||gamma fpr||^2
__device__ __forceinline__ bool panoc_lip_test_fails(const Ctx& cx, double cost_half, double cost, double ip, double ig, double nfpr) {
    const double rhs_lip = cost + KC(K_EPS_LIP) * fabs(cost) - ip + (KC(K_GAMMA_L) * 0.5 * ig) * nfpr * nfpr;
    return cost_half > rhs_lip;
}
// L <- 2L, gamma <- gamma / 2, new half step, fpr and the sums that belong to it
template <class P>
__device__ __forceinline__ void panoc_lip_update(const KParams& kp, bool vl, double uv, double uw, double gv, double gw, double& Lip,
                                                 double& gamma, double& ig, double& hv, double& hw, double& rv, double& rw, double& d2h,
                                                 double& nfpr, double& ip) {
    Lip = P::uni(Lip * 2.0); gamma = P::uni(gamma * 0.5); ig = P::uni(ig * 2.0);
    const double e2 = panoc_half_step(kp, vl, uv, uw, gamma, gv, gw, hv, hw);
    rv = uv - hv; rw = uw - hw;
    double rr;
    P::sum2(e2, __builtin_fma(rv, rv, rw * rw), d2h, rr);
    nfpr = P::uni(sqrt(rr));
    ip = dot2r<P, P::RV>(gv, gw, rv, rw);
}
// the inner tolerance of the running inner problem, where panoc_step_residual reads it (HD_AKKT)
__device__ __forceinline__ void set_inner_tolerance(const Ctx& cx, int lane, double akkt_tol) {
    if (lane == 0) *cx.akkt = akkt_tol;
    wave_sync();
}
// gamma*fpr of the step, its norm, <grad, gamma fpr>; true when the inner problem is solved: ||gamma fpr|| < eps and the
// AKKT residual || gfpr/gamma + grad - grad_prev || < eps_nu (grad_prev is the zero vector on the first step of an inner
// problem and the current gradient afterwards)
template <class P>
__device__ __forceinline__ bool panoc_step_residual(const Ctx& cx, const KParams& kp, bool vl, double uv, double uw, double hv, double hw,
                                                    double gv, double gw, double gamma, int iter, double akkt_tol, double& rv,
                                                    double& rw, double& nfpr, double& ip) {
    rv = uv - hv; rw = uw - hw;
    double rr;
    P::sum2(__builtin_fma(rv, rv, rw * rw), __builtin_fma(gv, rv, gw * rw), rr, ip);
    nfpr = P::uni(sqrt(rr));
    bool ex = nfpr < kp.tol;
    if (ex) {
        const double a0 = rv / gamma + (iter == 0 ? gv : 0.0), a1 = rw / gamma + (iter == 0 ? gw : 0.0);
        ex = sqrt(dot2r<P, P::RV>(a0, a1, a0, a1)) < *cx.akkt;   // == akkt_tol (set_inner_tolerance)
    }
    return ex;
}

// Where the L-BFGS state of one problem lives (per wavefront).  LM = [S; Y] as 2 mem rows of N (v, w) pairs, slot-major: in
// the problem's workspace record (L2-resident global memory) in the throughput kernel, in LDS in the latency kernel and in the
// L-BFGS-in-LDS build.  GG, LRHO stay in LDS for the whole solve; XA is scratch between two evaluations (it aliases the
// evaluation's own scratch: positions + stash).
struct LbMem {
    double* LM;     // [2 mem + 1][N][2]: rows 0..mem-1 = s of slot 0..mem-1, rows mem..2mem-1 = y, then one row of zeros
    double* LOLD;   // [N][4]: previous (u, gamma fpr)
    double* LRHO;   // [mem]
    double* LALPHA; // [mem]        two-loop form only
    double* GG;     // Gram form only: [mem][mem] s_i.y_j (slot indices), then y_i.y_j packed symmetric: entry (i >= j) at i(i+1)/2 + j
    double* XA;     // Gram form only: scratch, see PanocLbfgsGram
};

// L-BFGS buffer of PANOC [crate lbfgs: C-BFGS acceptance]: ring of `mem` pairs (s, y) = (delta u, delta gamma*fpr), newest at
// `head`; S, Y [mem][N][2], OLD [N][4] = previous (u, gamma*fpr), RHO [mem].
struct PanocLbfgs {
    int active = 0, head = 0;
    bool first = true;
    double hgamma = 1.0;  // s'y / y'y of the newest pair
    __device__ __forceinline__ void flush() { active = 0; first = true; }
    template <class P>
    __device__ __forceinline__ void update_ring(const Ctx& cx, bool vl, int lane, int N, int mem, double uv, double uw, double rv, double rw,
                                                double nfpr, double* LS, double* LY, double* LOLD, double* LRHO) {
        if (first) {
            first = false;
            if (vl) { LOLD[lane * 4] = uv; LOLD[lane * 4 + 1] = uw; LOLD[lane * 4 + 2] = rv; LOLD[lane * 4 + 3] = rw; }
            return;
        }
        double s0 = 0, s1 = 0, y0_ = 0, y1_ = 0;
        if (vl) {
            s0 = uv - LOLD[lane * 4]; s1 = uw - LOLD[lane * 4 + 1];
            y0_ = rv - LOLD[lane * 4 + 2]; y1_ = rw - LOLD[lane * 4 + 3];
        }
        double ys, ss;
        P::sum2(__builtin_fma(s0, y0_, s1 * y1_), __builtin_fma(s0, s0, s1 * s1), ys, ss);
        if (!(ss <= KC(K_DBLMIN) || ys <= KC(K_MIN_L)) && (ys > (KC(K_CBFGS) * nfpr) * ss)) {  // s'y / ||s||^2 > eps ||gamma fpr||, ||s||^2 > 0
            head = (head + mem - 1) % mem;
            if (vl) {
                LOLD[lane * 4] = uv; LOLD[lane * 4 + 1] = uw; LOLD[lane * 4 + 2] = rv; LOLD[lane * 4 + 3] = rw;
                LS[(head * N + lane) * 2] = s0; LS[(head * N + lane) * 2 + 1] = s1;
                LY[(head * N + lane) * 2] = y0_; LY[(head * N + lane) * 2 + 1] = y1_;
            }
            if (lane == 0) LRHO[head] = 1.0 / ys;
            hgamma = P::uni(ys / dot2r<P, P::RV>(y0_, y1_, y0_, y1_));
            active = (active + 1 < mem) ? active + 1 : mem;
        }
    }
    // d = H * (gamma*fpr): two-loop recursion, newest pair first.  LBG: S, Y live in the workspace record (global memory):
    // pair j+1 is requested before pair j is consumed.
    template <class P, bool LBG>
    __device__ __forceinline__ void direction_two_loop(bool vl, int lane, int N, int mem, double rv, double rw, const double* LS, const double* LY,
                                                       const double* LRHO, double* LALPHA, double& dv, double& dw) const {
        double q0 = rv, q1 = rw;
        if (LBG) {
            // The ring lives in the workspace record (L2, ~700 cycles away).  THREE pairs are kept in flight: pair p travels in
            // register set p % 3; a set is refilled (pair p + 3 on the way down the buffer, p - 3 on the way back) as soon as
            // its pair has been consumed, i.e. two recursion steps before it is needed again.  The second loop starts with the
            // three oldest pairs still in registers.  Same operations in the same order as the plain loops below.
            auto pair_at = [&](const double* base, int p) -> double2 {
                int sl = head + p;
                sl = sl >= mem ? sl - mem : sl;
                return vl ? *reinterpret_cast<const double2*>(base + (sl * N + lane) * 2) : make_double2(0.0, 0.0);
            };
            auto rho_at = [&](int p) -> double {
                int sl = head + p;
                sl = sl >= mem ? sl - mem : sl;
                return LRHO[sl];
            };
            const double2 z2 = make_double2(0.0, 0.0);
            double2 sA = z2, yA = z2, sB = z2, yB = z2, sC = z2, yC = z2;
            if (active > 0) { sA = pair_at(LS, 0); yA = pair_at(LY, 0); }
            if (active > 1) { sB = pair_at(LS, 1); yB = pair_at(LY, 1); }
            if (active > 2) { sC = pair_at(LS, 2); yC = pair_at(LY, 2); }
#define MPC_LB_DOWN(S_, Y_)                                                                          \
    {                                                                                                \
        const double al = rho_at(j) * dot2r<P, P::RV>(S_.x, S_.y, q0, q1);                           \
        if (lane == 0) LALPHA[j] = al;                                                               \
        q0 -= al * Y_.x; q1 -= al * Y_.y;                                                            \
        if (j + 3 < active) { S_ = pair_at(LS, j + 3); Y_ = pair_at(LY, j + 3); }                    \
        ++j;                                                                                         \
    }
            int j = 0;
            while (j < active) {
                MPC_LB_DOWN(sA, yA)
                if (j < active) MPC_LB_DOWN(sB, yB)
                if (j < active) MPC_LB_DOWN(sC, yC)
            }
#undef MPC_LB_DOWN
            wave_sync();
            if (active > 0) { q0 *= hgamma; q1 *= hgamma; }
#define MPC_LB_UP(S_, Y_)                                                                            \
    {                                                                                                \
        const double be = rho_at(j) * dot2r<P, P::RV>(Y_.x, Y_.y, q0, q1);                           \
        const double co = LALPHA[j] - be;                                                            \
        q0 += co * S_.x; q1 += co * S_.y;                                                            \
        if (j >= 3) { S_ = pair_at(LS, j - 3); Y_ = pair_at(LY, j - 3); }                            \
        --j;                                                                                         \
    }
            j = active - 1;
            int ph = j >= 0 ? j % 3 : 0;  // register set of the oldest pair
            while (j >= 0) {
                if (ph == 2) { MPC_LB_UP(sC, yC) ph = 1; if (j < 0) break; }
                if (ph == 1) { MPC_LB_UP(sB, yB) ph = 0; if (j < 0) break; }
                MPC_LB_UP(sA, yA) ph = 2;
            }
#undef MPC_LB_UP
        } else {
            for (int j = 0; j < active; ++j) {
                const int sl = (head + j) % mem;
                const double sj0 = vl ? LS[(sl * N + lane) * 2] : 0.0, sj1 = vl ? LS[(sl * N + lane) * 2 + 1] : 0.0;
                const double al = LRHO[sl] * dot2r<P, P::RV>(sj0, sj1, q0, q1);
                if (lane == 0) LALPHA[j] = al;
                if (vl) { q0 -= al * LY[(sl * N + lane) * 2]; q1 -= al * LY[(sl * N + lane) * 2 + 1]; }
            }
            wave_sync();
            if (active > 0) { q0 *= hgamma; q1 *= hgamma; }
            for (int j = active - 1; j >= 0; --j) {
                const int sl = (head + j) % mem;
                const double yj0 = vl ? LY[(sl * N + lane) * 2] : 0.0, yj1 = vl ? LY[(sl * N + lane) * 2 + 1] : 0.0;
                const double be = LRHO[sl] * dot2r<P, P::RV>(yj0, yj1, q0, q1);
                const double co = LALPHA[j] - be;
                if (vl) { q0 += co * LS[(sl * N + lane) * 2]; q1 += co * LS[(sl * N + lane) * 2 + 1]; }
            }
        }
        dv = q0; dw = q1;
    }
    // the interface the solver kernels use (shared with PanocLbfgsGram)
    template <class P, int NT, int MEMT>
    __device__ __forceinline__ void update(const Ctx& cx, const KParams& kp, bool vl, int lane, double uv, double uw, double rv,
                                           double rw, double nfpr, const LbMem& m, double&) {
        const int N = NT ? NT : kp.N;
        update_ring<P>(cx, vl, lane, N, kp.mem, uv, uw, rv, rw, nfpr, m.LM, m.LM + kp.mem * N * 2, m.LOLD, m.LRHO);
    }
    template <class P, int NT, int MEMT, bool LBG>
    __device__ __forceinline__ void direction(const Ctx& cx, const KParams& kp, bool vl, int lane, double rv, double rw, const LbMem& m,
                                              double, double& dv, double& dw) const {
        const int N = NT ? NT : kp.N;
        direction_two_loop<P, LBG>(vl, lane, N, kp.mem, rv, rw, m.LM, m.LM + kp.mem * N * 2, m.LRHO, m.LALPHA, dv, dw);
    }
};

// The same buffer and the same operator H as PanocLbfgs, evaluated in GRAM FORM (round 3; the CPU checker restates it as lbfgs_apply_gram).
// The two-loop recursion is 2 mem DEPENDENT wave reductions, each waiting for a pair from L2 and using N of 64 lanes.  Here:
//   pass 1  one matrix-vector pass [S; Y] x over all 64 lanes: lane (row, g) accumulates its share of row . x for
//           x = gamma fpr and x = y_new (independent FMAs, no reduction; the G partials of a row sit in adjacent lanes);
//   Gram    the products with y_new are the new column of the cached matrices s_i.y_j, y_i.y_j (LDS);
//   loops   both loops of the recursion become scalar recurrences on those matrices: the lane of a row carries its own running
//           product, one v_readlane broadcasts the coefficient of the step, one FMA updates all rows;
//   pass 2  d = gamma_H r + sum_rows coef[row] * row, again over all lanes (rows split in G2 groups, combined through LDS).
// Lanes: the s-row of slot l lives in lanes l*G .. l*G+G-1, its y-row 32 lanes higher; a row's total ends in its last lane.
// Of s_i.y_j only the entries with s_i OLDER than y_j are ever used (first loop: s_p.q_p needs y_q of the newer pairs; second
// loop: y_p.z needs s_q of the older pairs).  The others -- and the diagonal, which only enters through rho -- are stored as
// ZERO, so a row's running product stops changing by itself once its turn has passed: no per-step masking or latching.
// Exact-arithmetic identical to the two-loop recursion; measured against it over whole solves: 5e-12 relative (oracle, mode 2).
constexpr int LB_SCHED = 2;   // pass 1: the scheduler may not pull the LDS operands of more than this many pairs ahead (registers)
constexpr int LB_BLOCK = 8;   // pass 1: rows held in registers at a time
// (both passes request their rows before anything that depends on the new pair / the recurrences: measured in round 3)
struct PanocLbfgsGram {
    int active = 0, head = 0;
    bool first = true;
    double hgamma = 1.0;
    __device__ __forceinline__ void flush() { active = 0; first = true; }

    static __device__ __forceinline__ int yy_index(int i, int j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }
    template <int NT, int MEMT>
    struct Dims {
        int N, mem, R, G, CL, G2, CR;
        __device__ __forceinline__ Dims(const KParams& kp) {
            N = NT ? NT : kp.N; mem = MEMT ? MEMT : kp.mem;
            R = 2 * mem; G = (WAVE / 2) / mem; CL = (N + G - 1) / G;   // G >= 2 (mem <= 16): chunks of CL (v, w) pairs
            G2 = WAVE / N; CR = (R + G2 - 1) / G2;                     // pass 2: G2 groups of CR rows
        }
        __device__ __forceinline__ int xa_len() const { return G * CL * 4; }  // (r, y) pairs of every chunk slot, zero padded
    };
    // lane -> (y-row?, slot, chunk); lanes whose slot is >= mem compute something that is never read
    struct Role { int isy, slot, slot_a, g; };
    template <int NT, int MEMT>
    static __device__ __forceinline__ Role role(const Dims<NT, MEMT>& D, int lane) {
        Role r;
        r.isy = lane >> 5;
        const int l = lane & 31;
        r.slot = l / D.G; r.g = l - r.slot * D.G;
        r.slot_a = r.slot < D.mem ? r.slot : D.mem - 1;
        return r;
    }

    // `pr` (out): row lanes: (row of [S; Y]) . (gamma fpr) -- what direction() starts from; it lives only between the two calls
    template <class P, int NT, int MEMT>
    __device__ __forceinline__ void update(const Ctx& cx, const KParams& kp, bool vl, int lane, double uv, double uw, double rv,
                                           double rw, double nfpr, const LbMem& m, double& pr) {
        const Dims<NT, MEMT> D(kp);
        const int N = D.N, mem = D.mem;
        double* LOLD = m.LOLD;
        if (first) {
            first = false;
            if (vl) { LOLD[lane * 4] = uv; LOLD[lane * 4 + 1] = uw; LOLD[lane * 4 + 2] = rv; LOLD[lane * 4 + 3] = rw; }
            return;
        }
        // pass-1 operands of this lane: its row of [S; Y], chunk g of the row's (v, w) pairs -- CL consecutive pairs (the last
        // chunk runs into the next row, or into the zero row that follows the last one: a finite value times the zero padding
        // of XA).  The OLD rows do not depend on the new pair: with a compile-time shape they are requested before anything else.
        asm volatile("" : "+v"(lane));  // opaque: the per-lane addresses are rebuilt here, not hoisted out of the solver loop (and spilled)
        const Role ro = role(D, lane);
        const int row_a = ro.isy * mem + ro.slot_a;
        const double2* mrow = reinterpret_cast<const double2*>(m.LM + (row_a * N + ro.g * D.CL) * 2);
        constexpr int GT = MEMT ? (WAVE / 2) / (MEMT ? MEMT : 1) : 1;
        constexpr int CLT = (NT && MEMT) ? (NT + GT - 1) / GT : 0;   // pairs per lane (compile-time shape)
        constexpr int BL = CLT <= LB_BLOCK ? CLT : (CLT + 1) / 2;   // held in registers at a time (else two blocks)
        double2 mq[BL ? BL : 1];
        if (CLT) {
#pragma unroll
            for (int t = 0; t < BL; ++t) mq[t] = mrow[t];
        }
        double s0 = 0, s1 = 0, y0_ = 0, y1_ = 0;
        if (vl) {
            s0 = uv - LOLD[lane * 4]; s1 = uw - LOLD[lane * 4 + 1];
            y0_ = rv - LOLD[lane * 4 + 2]; y1_ = rw - LOLD[lane * 4 + 3];
        }
        for (int k = lane; k < D.G * D.CL; k += WAVE) {  // slots N .. G*CL-1 pad the last chunk with zeros
            double2* x = reinterpret_cast<double2*>(m.XA + k * 4);
            const bool real = vl && k == lane;
            x[0] = make_double2(real ? rv : 0.0, real ? rw : 0.0); x[1] = make_double2(real ? y0_ : 0.0, real ? y1_ : 0.0);
        }
        double ys, ss;
        P::sum2(__builtin_fma(s0, y0_, s1 * y1_), __builtin_fma(s0, s0, s1 * s1), ys, ss);
#if MPC_LB_HOIST_SUMS
        // the three sums of an ACCEPTED pair (y'y, s'r, y'r), formed here -- next to the two of the acceptance test and ahead of pass 1 -- instead of
        // behind the test: three independent reduction chains the scheduler can interleave (a lone wavefront of the latency kernel waits out
        // every DPP step of a single chain); a rejected pair (rare) has computed them for nothing.  The same sums: same bits.
        double yy_h, sr_h;
        P::sum2(__builtin_fma(y0_, y0_, y1_ * y1_), __builtin_fma(s0, rv, s1 * rw), yy_h, sr_h);
        const double yr_h = dot2r<P, P::RV>(y0_, y1_, rv, rw);
#endif
        wave_sync();
        double ar = 0.0, ay = 0.0;
        auto mac = [&](const double2 mv, int k) {
            const double2* x = reinterpret_cast<const double2*>(m.XA + k * 4);
            const double2 xr = x[0], xy = x[1];
            ar = __builtin_fma(mv.x, xr.x, ar); ar = __builtin_fma(mv.y, xr.y, ar);
            ay = __builtin_fma(mv.x, xy.x, ay); ay = __builtin_fma(mv.y, xy.y, ay);
        };
        if (CLT) {
#pragma unroll
            for (int t = 0; t < BL; ++t) {
                mac(mq[t], ro.g * CLT + t);
                if ((t % LB_SCHED) == LB_SCHED - 1) __builtin_amdgcn_sched_barrier(0);
            }
            if (BL < CLT) {
#pragma unroll
                for (int t = 0; t < CLT - BL; ++t) mq[t] = mrow[BL + t];
#pragma unroll
                for (int t = 0; t < CLT - BL; ++t) mac(mq[t], ro.g * CLT + BL + t);
            }
        } else {
            for (int t = 0; t < D.CL; ++t) mac(mrow[t], ro.g * D.CL + t);
        }
        {   // the G partials of a row are in adjacent lanes: ((g0 + g1) + g2) ends in the row's last lane
            double tr = ar, ty = ay;
            for (int i = 1; i < D.G; ++i) { tr = ar + dpp0<DPP_WAVE_SHR1>(tr); ty = ay + dpp0<DPP_WAVE_SHR1>(ty); }
            ar = tr; ay = ty;
        }
        pr = ar;
        if (!(ss <= KC(K_DBLMIN) || ys <= KC(K_MIN_L)) && (ys > (KC(K_CBFGS) * nfpr) * ss)) {  // s'y / ||s||^2 > eps ||gamma fpr||, ||s||^2 > 0
#if MPC_LB_HOIST_SUMS
            const double yy = yy_h, sr = sr_h, yr = yr_h;
#else
            double yy, sr;
            P::sum2(__builtin_fma(y0_, y0_, y1_ * y1_), __builtin_fma(s0, rv, s1 * rw), yy, sr);
            const double yr = dot2r<P, P::RV>(y0_, y1_, rv, rw);
#endif
            head = head == 0 ? mem - 1 : head - 1;
            const int h = head;
            if (vl) {
                LOLD[lane * 4] = uv; LOLD[lane * 4 + 1] = uw; LOLD[lane * 4 + 2] = rv; LOLD[lane * 4 + 3] = rw;
                *reinterpret_cast<double2*>(m.LM + (h * N + lane) * 2) = make_double2(s0, s1);
                *reinterpret_cast<double2*>(m.LM + ((mem + h) * N + lane) * 2) = make_double2(y0_, y1_);
            }
            // Column h of both Gram matrices from the totals of pass 1; row h of s_i.y_j is zero (s_new is newer than every y),
            // and so is its diagonal.  The rows of slot h itself held the pair that is being replaced: what they contribute is
            // overwritten right after -- the LDS writes of a wavefront keep their order.
            if (ro.slot < mem && ro.g == D.G - 1) {
                if (!ro.isy) {
                    m.GG[ro.slot * mem + h] = ay;                  // s_q . y_new
                    if (ro.slot == h) pr = sr;
                } else {
                    m.GG[mem * mem + yy_index(ro.slot, h)] = ay;   // y_q . y_new (symmetric: stored once)
                    if (ro.slot == h) pr = yr;
                }
            }
            wave_sync();
            if (lane < mem) m.GG[h * mem + lane] = zero_here();   // (a hoisted 0.0 sat in a spill slot and came back from scratch in every step)
            if (lane == 0) {
                m.GG[mem * mem + yy_index(h, h)] = yy;
                m.LRHO[h] = 1.0 / ys;
            }
            hgamma = P::uni(ys / yy);
            active = (active + 1 < mem) ? active + 1 : mem;
        }
    }

    template <class P, int NT, int MEMT, bool LBG>
    __device__ __forceinline__ void direction(const Ctx& cx, const KParams& kp, bool vl, int lane, double rv, double rw, const LbMem& m,
                                              double pr, double& dv, double& dw) const {
        if (active == 0) { dv = rv; dw = rw; return; }
        const Dims<NT, MEMT> D(kp);
        const int N = D.N, mem = D.mem;
        asm volatile("" : "+v"(lane));  // opaque, see update()
        const Role ro = role(D, lane);
        double* COEF = m.XA + D.xa_len();  // [G2 * CR] coefficients of the rows (zero beyond the active pairs)
        // pass-2 operands: lane = g2 * N + k2 takes rows g2 * CR .. (the row after the last one is the zero row); they depend on
        // nothing computed here, so with a compile-time shape they are requested before the recurrences
        const int g2 = lane / N, k2 = lane - g2 * N;
        const int g2a = g2 < D.G2 ? g2 : D.G2 - 1;
        constexpr int CRT0 = (NT && MEMT) ? (2 * MEMT + WAVE / (NT ? NT : 1) - 1) / (WAVE / (NT ? NT : 1)) : 0;
        constexpr int CRT = CRT0 <= 8 ? CRT0 : 0;  // long horizons: one lane group takes all rows, too many to hold in registers
        {   // zero materialised here: hoisted out of the solver loop it would sit in a spill slot, and its reload would wait for
            // the row loads below (scratch and global loads share one counter)
            double z = 0.0;
            asm volatile("" : "+v"(z));
            if (lane < D.G2 * D.CR) COEF[lane] = z;
        }
        const double rho_l = m.LRHO[ro.slot_a];
        const int tl = ro.slot_a * D.G + D.G - 1;  // lane that holds the total of this slot's s-row
        double2 mq[CRT ? CRT : 1];
        const double2* mcol = reinterpret_cast<const double2*>(m.LM + (g2a * D.CR * N + k2) * 2);  // + tt * N: row g2a * CR + tt
        if (CRT) {
#pragma unroll
            for (int tt = 0; tt < CRT; ++tt) mq[tt] = mcol[tt * N];
        }
        // first loop, newest pair first: alpha_p = rho_p s_p.q_p; every row's product with q advances by -alpha_p (row . y_p)
        // s-rows read row slot of s_i.y_j; y-rows read entry (slot, sp) of the packed symmetric y_i.y_j
        const double* gs = m.GG + ro.slot_a * mem;
        const double* gy = m.GG + mem * mem;
        const int tri_l = ro.slot_a * (ro.slot_a + 1) / 2;
        auto g1 = [&](int sp) { return ro.isy ? gy[sp <= ro.slot_a ? tri_l + sp : sp * (sp + 1) / 2 + ro.slot_a] : gs[sp]; };
        double acc = pr;
        {
            int sp = head;
            double gnext = g1(sp);
            for (int p = 0; p < active; ++p) {
                const double gcur = gnext;
                const int sl = sp * D.G + D.G - 1;
                sp = sp + 1 >= mem ? 0 : sp + 1;
                gnext = g1(sp);   // one step ahead of the broadcast that needs it
                const double al = readlane_d(rho_l * acc, sl);
                acc = __builtin_fma(-al, gcur, acc);
            }
        }
        // alpha of every slot: final in the lane of its s-row (the later steps multiplied zeros); the y-rows get a copy
        double al_l = rho_l * acc;
        {
            const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(al_l), (unsigned)__double2loint(al_l), false, false);
            const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(al_l), (unsigned)__double2hiint(al_l), false, false);
            al_l = __hiloint2double((int)hi[0], (int)lo[0]);  // lanes 0..31 keep theirs, lanes 32..63 receive that of lanes 0..31
        }
        // second loop, oldest pair first: beta_p = rho_p y_p.z; the y-rows' products with z advance by delta_p (s_p . y_row)
        double t = hgamma * acc;
        {
            int sp = head + active - 1;
            sp = __builtin_amdgcn_readfirstlane(sp >= mem ? sp - mem : sp);
            const double* gcol = m.GG + ro.slot_a;
            double gnext = gcol[sp * mem];
            for (int p = active - 1; p >= 0; --p) {
                const double gcur = gnext;
                const int sl = 32 + sp * D.G + D.G - 1;
                sp = __builtin_amdgcn_readfirstlane(sp == 0 ? mem - 1 : sp - 1);
                gnext = gcol[sp * mem];
                const double de = readlane_d(__builtin_fma(-rho_l, t, al_l), sl);
                t = __builtin_fma(de, gcur, t);
            }
        }
        // coefficients of the rows: delta_l for s_l, -gamma_H alpha_l for y_l (slots without a pair keep 0)
        {
            int pos = ro.slot - head;
            pos = pos < 0 ? pos + mem : pos;
            if (ro.isy && ro.slot < mem && ro.g == D.G - 1 && pos < active) {
                COEF[ro.slot] = __builtin_fma(-rho_l, t, al_l);
                COEF[mem + ro.slot] = -(hgamma * al_l);
            }
        }
        wave_sync();
        double a0 = 0.0, a1 = 0.0;
        if (CRT) {
#pragma unroll
            for (int tt = 0; tt < CRT; ++tt) {
                const double cf = COEF[g2a * CRT + tt];
                a0 = __builtin_fma(cf, mq[tt].x, a0); a1 = __builtin_fma(cf, mq[tt].y, a1);
            }
        } else {
            for (int tt = 0; tt < D.CR; ++tt) {
                const int r2 = g2a * D.CR + tt;
                const double2 mv = mcol[(r2 <= D.R ? tt : 0) * N];  // rows beyond the zero row do not exist: their coefficient is 0
                const double cf = COEF[r2];
                a0 = __builtin_fma(cf, mv.x, a0); a1 = __builtin_fma(cf, mv.y, a1);
            }
        }
        // the partials of groups 1.. travel through the scratch that held the pass-1 operands
        double2* P2 = reinterpret_cast<double2*>(m.XA);
        if (g2 >= 1 && g2 < D.G2) P2[(g2 - 1) * N + k2] = make_double2(a0, a1);
        wave_sync();
        double d0 = 0.0, d1 = 0.0;
        if (vl) {
            d0 = __builtin_fma(hgamma, rv, a0); d1 = __builtin_fma(hgamma, rw, a1);
            for (int i = 1; i < D.G2; ++i) { const double2 q = P2[(i - 1) * N + lane]; d0 += q.x; d1 += q.y; }
        }
        dv = d0; dw = d1;
    }
};

// Where the Gram form is used.  N_hor = 20: -7 % against the two-loop recursion (profiles/archive/r03_lbfgs_gram_ab.txt).  N_hor = 40: the
// first measurement said +7 % -- the 1.2 KB of Gram matrices had cost a resident wavefront per CU (LDS comes in 1280-byte granules);
// at EQUAL residency it is 4 % faster there too, and with the stash-free carve (stash_stride_c: 12 368 B = 12 per CU) the kernel
// went from 954 to 913 ms at B = 16 384.  One lane group does all 2 mem rows in pass 2 there (64 / N = 1).  The runtime-horizon
// kernel keeps the two-loop form.
template <int NT> struct GramFor { static constexpr bool value = MPC_LBFGS_GRAM && NT != 0 && WAVE / (NT ? NT : 1) >= 2; };
template <bool DUO, int NT> struct LbfgsOf { using type = PanocLbfgs; };
template <> struct LbfgsOf<false, 40> { using type = std::conditional<gram_shape(40, 10), PanocLbfgsGram, PanocLbfgs>::type; };
template <> struct LbfgsOf<false, 20> { using type = std::conditional<GramFor<20>::value, PanocLbfgsGram, PanocLbfgs>::type; };

// ------------------------------------------------------------------------------------------------
// test-hook kernel: the L-BFGS operator alone.  A recorded sequence of iterates and residuals (u_j, gamma fpr_j), j = 0 .. m, is
// fed to the buffer exactly as solve_body feeds it (update, then the direction from the last residual) -- once through the
// Gram form, once through the two-loop recursion -- and both directions come back (tests/test_gpu_baseline_parity.py compares
// them with each other and with a host restatement of the recursion of the `lbfgs` crate).
// ------------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(WAVE) void lbfgs_direction_kernel(KParams kp, double* __restrict__ wsb, const double* __restrict__ U,
                                                               const double* __restrict__ R, int m, double* d_gram, double* d_two,
                                                               int32_t* pairs, int B) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    using P = Solo<NT>;
    const int b = blockIdx.x;
    if (b >= B) return;
    const int lane = threadIdx.x, N = NT, mem = kp.mem;
    constexpr int MEMT = MemOf<NT>::value;
    constexpr FixedLds FL = fixed_lds(NT, MEMT, false);
    Ctx cx{};
    double* hd = lds + FL.hd;
    if (lane < 27) hd[KC_BASE + lane] = KTAB[lane];
    cx.hd = hd; cx.pos = lds + FL.pos; cx.lane = lane; cx.vl = lane < N;
    LbMem lm;
    lm.LM = wsb + (size_t)b * kp.ws_stride + kp.ws_lbs;
    lm.LOLD = wsb + (size_t)b * kp.ws_stride + kp.ws_lold;
    lm.LRHO = lds + FL.rho; lm.LALPHA = lds + FL.gg; lm.GG = lds + FL.gg; lm.XA = cx.pos;
    const bool vl = lane < N;
    auto run = [&](auto& lb, double* out, int which) {
        for (int i = lane; i < (2 * mem + 1) * N; i += WAVE) reinterpret_cast<double2*>(lm.LM)[i] = make_double2(0.0, 0.0);
        wave_sync();
        double dv = 0.0, dw = 0.0;
        for (int j = 0; j <= m; ++j) {
            const double* uj = U + ((size_t)b * (m + 1) + j) * 2 * N;
            const double* rj = R + ((size_t)b * (m + 1) + j) * 2 * N;
            const double uv = vl ? uj[2 * lane] : 0.0, uw = vl ? uj[2 * lane + 1] : 0.0;
            const double rv = vl ? rj[2 * lane] : 0.0, rw = vl ? rj[2 * lane + 1] : 0.0;
            const double nfpr = P::uni(sqrt(dot2r<P, P::RV>(rv, rw, rv, rw)));
            double pr = 0.0;
            lb.template update<P, NT, MEMT>(cx, kp, vl, lane, uv, uw, rv, rw, nfpr, lm, pr);
            wave_sync();
            if (j == m) lb.template direction<P, NT, MEMT, true>(cx, kp, vl, lane, rv, rw, lm, pr, dv, dw);
        }
        if (vl) { out[(size_t)b * 2 * N + 2 * lane] = dv; out[(size_t)b * 2 * N + 2 * lane + 1] = dw; }
        if (lane == 0 && pairs) pairs[2 * b + which] = lb.active;
        wave_sync();
    };
    PanocLbfgsGram g;
    run(g, d_gram, 0);
    PanocLbfgs t;
    run(t, d_two, 1);
}

// ALM / PM outer step: y+ <- y + c (F1(u) - Proj_C(F1(u) + y/c)); ||y+ - y||
template <class P>
__device__ __forceinline__ void alm_multiplier_step(const KParams& kp, bool vl, double F1a, double F1b, double ya, double yb, double c,
                                                    double& ypa, double& ypb, double& dy_norm_plus) {
    double dy2l = 0.0;
    ypa = 0.0; ypb = 0.0;
    if (vl) {
        const double za = F1a + ya / c, zb = F1b + yb / c;
        ypa = ya + c * (F1a - clampd(za, kp.amin, kp.amax));
        ypb = yb + c * (F1b - clampd(zb, -kp.aamax, kp.aamax));
        dy2l = __builtin_fma(ypa - ya, ypa - ya, (ypb - yb) * (ypb - yb));
    }
    dy_norm_plus = P::uni(sqrt(P::template sum<P::RV>(dy2l)));
}
__device__ __forceinline__ bool alm_exit(const Ctx& cx, const KParams& kp, int alm_iteration, double dy_norm_plus, double f2_norm_plus,
                                         double akkt_tol, double c) {
    const bool crit1 = alm_iteration > 0 && dy_norm_plus <= c * kp.delta_tol + KC(K_EPS);
    const bool crit2 = f2_norm_plus <= kp.delta_tol + KC(K_EPS);
    const bool crit3 = akkt_tol <= kp.tol + KC(K_EPS);
    return crit1 && crit2 && crit3;
}
// The penalty stays ("stall criterion") in the first outer iteration and when the infeasibility shrank by the factor theta.  Two
// readings [OpEn; cannot be checked against the crate here, DESIGN.md section 3], option MPCGPU_OPT_PENALTY_STALL:
//   stall_rule = 0 "either" (default): ||y+ - y|| OR ||F2|| shrank -- `is_penalty_stall_criterion` of the published engine as
//       recalled: iteration == 0 || (n1 > 0 && dy+ <= theta dy + eps) || (n2 > 0 && ||F2+|| <= theta ||F2|| + eps).  With inactive
//       acceleration constraints y+ = y = 0, so 0 <= theta * 0 + eps holds and the penalty keeps its initial value 10.
//   stall_rule = 1 "both": both shrank (rounds 1-5 of this build).
// n1 = 2 N_hor > 0 and n2 = Ndynobs > 0 for every configuration the library accepts.
__device__ __forceinline__ bool alm_stalled(const Ctx& cx, const KParams& kp, int alm_iteration, double dy_norm_plus, double dy_norm,
                                            double f2_norm_plus, double f2_norm) {
    const bool alm_shrank = dy_norm_plus <= kp.suff_decrease * dy_norm + KC(K_EPS);
    const bool pm_shrank = f2_norm_plus <= kp.suff_decrease * f2_norm + KC(K_EPS);
    return alm_iteration == 0 || (kp.stall_rule == 1 ? (alm_shrank && pm_shrank) : (alm_shrank || pm_shrank));
}

// The whole ALM / PANOC solve of problem P::problem() on the lanes P gives it.  `lds` is the workgroup's dynamic LDS.
// The rare evaluations are states of a small machine; the PANOC steps run in a loop of their own (MPC_STEP_LOOP).
template <int NT, bool SC, bool LBG, class P, bool AXIS = false, bool LIN = false, int MINW = 3>
__device__ __forceinline__ void solve_body(const KParams& kp, const BatchPtrs& io, int B, double* lds) {
    if (P::problem() >= B) return;
    if (io.nsel && P::problem() >= *io.nsel) return;               // pick-up launch: only the listed problems
    const int b = io.perm ? io.perm[P::problem()] : P::problem();   // every output below is indexed by the PROBLEM, not by the workgroup
    const long long t_start = wall_clock64();
    const int lane = P::lane(), N = NT ? NT : kp.N, mem = kp.mem;
    lds += P::half() * kp.l_total;  // this problem's carve
    const double* ws = io.ws + (size_t)b * kp.ws_stride;
    // a problem whose dynamic rows do not fit the linear centre tables is left to the pick-up launch that follows (stored centres)
    if (LIN && P::uni(ws[H_NLIN]) != 0.0) return;
    // Launches with a RESERVED LDS carve (mpcgpu_reserve_shape: no count read-back before the launch) check every problem
    // against it: a problem with more active rows than reserved must not touch the tables -- it is reported, not solved.
    if (kp.reserved) {
        const bool over = (int)P::uni(ws[H_KS]) > kp.mKs || (int)P::uni(ws[H_KF]) > kp.mKf || (int)P::uni(ws[H_KD]) > kp.mKd ||
                          (SC && P::uni(ws[H_VAR]) != 0.0) || (AXIS && P::uni(ws[H_ROT]) != 0.0);
        if (over) {
            const double nan = __builtin_nan("");
            if (lane < N) {
                io.u[(size_t)b * 2 * N + 2 * lane] = 0.0; io.u[(size_t)b * 2 * N + 2 * lane + 1] = 0.0;
                if (io.y_out) { io.y_out[(size_t)b * 2 * N + lane] = nan; io.y_out[(size_t)b * 2 * N + N + lane] = nan; }
            }
            if (lane == 0) {
                io.cost[b] = nan; io.status[b] = 4;
                if (io.inner_it) io.inner_it[b] = 0;
                if (io.outer_it) io.outer_it[b] = 0;
                if (io.evals) { io.evals[2 * b] = 0; io.evals[2 * b + 1] = 0; }
                if (io.fpr) io.fpr[b] = nan;         // every optional output is written: no stale value of an earlier call survives
                if (io.f2norm) io.f2norm[b] = nan;
                if (io.ms) io.ms[b] = 0.0;
                if (kp.yield_from > 0) {   // begun and finished at once (the counters of the tail promotion: every problem is in both)
                    __hip_atomic_fetch_add(io.counts + CNT_STARTED, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_fetch_add(io.counts + CNT_FINISHED, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            return;
        }
    }
    Ctx cx;
    load_problem<NT, SC, P, LIN, MINW>(kp, ws, lds, cx);
    // L-BFGS memory S, Y [mem][N][2]: in LDS (LBG = false) or in this problem's workspace record, i.e. in the
    // L2-resident HBM workspace (LBG = true: 6.4 KB less LDS per wavefront -> more resident wavefronts; the pairs are
    // streamed once per PANOC iteration, one pair ahead of the dot product that consumes them)
    constexpr int MEMT = MemOf<NT>::value;
    LbMem lm;
    constexpr bool FIXED = NT != 0 && !P::DUO;   // compile-time offsets of the fixed part of the carve (fixed_lds)
    constexpr FixedLds FL = fixed_lds(NT ? NT : 2, MEMT ? MEMT : 1, !LBG, MINW);
    lm.LM = LBG ? io.ws + (size_t)b * kp.ws_stride + kp.ws_lbs : lds + (FIXED ? FL.S : kp.l_S);  // [S; Y], contiguous in both layouts
    lm.LRHO = lds + (FIXED ? FL.rho : kp.l_rho);  // [mem]
    lm.LALPHA = lds + (FIXED ? FL.gg : kp.l_alpha);
    // [N][4]: L-BFGS old state (u) and old g (gamma*fpr); read and written once per PANOC iteration by its own lane.
    // It follows S and Y into the workspace record (measured: keeping it in LDS when it still fits is no faster for the
    // benchmark batch and slower for small batches and for N = 40).
    lm.LOLD = LBG ? io.ws + (size_t)b * kp.ws_stride + kp.ws_lold : lds + (FIXED ? FL.old : kp.l_old);
    lm.GG = lds + (FIXED ? FL.gg : kp.l_gg);
    lm.XA = cx.pos;  // scratch between two evaluations: positions + stash are dead there
    // pass 2 of the Gram form multiplies EVERY row by its coefficient (0 for the slots that hold no pair): the rows must be finite
    for (int i = lane; i < (2 * mem + 1) * N; i += P::W) reinterpret_cast<double2*>(lm.LM)[i] = make_double2(0.0, 0.0);
    const bool vl = cx.vl;
    const int MAX_LIP_IT = 10, MAX_LS_IT = 10;

    // decision vector and multipliers (vector lanes; zeros elsewhere)
    double u0v = 0.0, u0w = 0.0, ya = 0.0, yb = 0.0;
    if (vl) {
        if (io.u0) { u0v = io.u0[(size_t)b * 2 * N + 2 * lane]; u0w = io.u0[(size_t)b * 2 * N + 2 * lane + 1]; }
        if (io.y0) { ya = io.y0[(size_t)b * 2 * N + lane]; yb = io.y0[(size_t)b * 2 * N + N + lane]; }
    }
    double uv = u0v, uw = u0w;
    double c = kp.init_penalty;
    if (io.c0) { const double c0 = io.c0[b]; if (c0 > 0.0) c = c0; }
    c = P::uni(c);
    double icm = P::uni(1.0 / fmax(c, 1.0));  // 1 / max(c, 1) of the augmented Lagrangian: changes with the penalty only
    ya = clampd(ya, -KC(K_YBOUND), KC(K_YBOUND)); yb = clampd(yb, -KC(K_YBOUND), KC(K_YBOUND));  // y <- Proj_Y(y), Y = [-1e12, 1e12]^n1
    // PANOC cache: vector state u, grad, u_half, gamma*fpr, direction (2 doubles per vector lane each);
    // ||grad||^2 and ||gradient_step - u_half||^2 are carried as scalars (they only enter the envelope)
    double gv = 0, gw = 0, hv = 0, hw = 0, rv_ = 0, rw_ = 0, dv = 0, dw = 0;
    double gamma = 0, ig = 0, Lip = 0, sigma = 0, cost = 0, nfpr = 0, tau = 1, rhs = 0, nh = 1, gg = 0, d2h = 0;
    double ip = 0.0;  // <grad, gamma*fpr> of the current step (Lipschitz test)
    double akkt_tol = kp.init_tol;
    set_inner_tolerance(cx, lane, akkt_tol);
    int iter = 0, num_iter = 0, lip_it = 0, nls = 0;
    bool cont_iters = true, cont_time = true;
    typename LbfgsOf<P::DUO, NT>::type lb;
    // ALM cache
    int alm_iteration = 0, num_outer = 1, inner_total = 0, status = 0;
    double dy_norm = 0, dy_norm_plus = 0, f2_norm = 0, f2_norm_plus = 0, last_fpr = 0, f_final = 0;

    int n_eval = 0, n_eval_grad = 0;
#ifdef MPC_TRACE
    int tr_n = 0;
    double tr_psi_u = 0.0;
    auto tr_write = [&](int nls_, double tau_) {
        if (lane == 0 && io.trace && tr_n < io.trace_cap) {
            double* r = io.trace + ((size_t)b * io.trace_cap + tr_n) * TRACE_W;
            r[0] = alm_iteration; r[1] = iter; r[2] = c; r[3] = Lip; r[4] = gamma; r[5] = nfpr; r[6] = tr_psi_u;
            r[7] = lip_it; r[8] = lb.active; r[9] = nls_; r[10] = tau_; r[11] = cost;
        }
        ++tr_n;
    };
#endif
    int state = ST_INIT0;
    double ev = uv, ew = uw;  // evaluation point
    bool want_grad = true;
    EvalOut o;
#ifdef MPC_PROFILE
    Prof prof; prof.start();
#endif
    // tail promotion (see YIELD at KParams): this problem leaves at a step boundary once the launch is draining
    constexpr bool CAN_YIELD = MPC_STEP_LOOP && !P::DUO && !LIN;
    [[maybe_unused]] bool yielded = false, count_step = false;
    int yslot = 0;

    for (;;) {
        if (CAN_YIELD && state == ST_INIT0 && kp.yield_from > 0) {
            // Start of an inner problem: the launch is draining when all but yield_cap of its problems have finished -- hand this one
            // to the latency kernel.  Here the state of the iteration is small (point, multipliers, penalty, tolerance, counters: the
            // PANOC cache and the L-BFGS buffer are empty) and nothing is added to the loop of the PANOC steps.
            // (first look of this problem: it counts itself as started -- the gate of the concurrent continuation waits until every
            //  problem of the launch has, mpc_team.hpp tail_gate_kernel)
            if (num_outer == 1 && lane == 0) __hip_atomic_fetch_add(io.counts + CNT_STARTED, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int fin_now = __hip_atomic_load(io.counts + CNT_FINISHED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bool leave = fin_now >= kp.yield_from;
            if (!leave && kp.yield_grad > 0) {   // gradual promotion: see KParams::yield_grad
                const int begun = __hip_atomic_load(io.counts + CNT_STARTED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int gone = __hip_atomic_load(io.counts + CNT_YIELDED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                leave = begun >= kp.yield_total && 2 * gone < kp.yield_cap && (long long)(gone + 1) * kp.yield_grad <= fin_now;
            }
            if (leave) {
                if (lane == 0) yslot = __hip_atomic_fetch_add(io.counts + CNT_YIELDED, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                yslot = __builtin_amdgcn_readfirstlane(yslot);
                if (yslot < kp.yield_cap) { yielded = true; break; }   // (the list cannot overflow: at most yield_cap problems are unfinished)
            }
        }
        PROF_MARK(10 + state);  // solver logic that led to this evaluation (by the state it was issued for)
        PROF_COUNT(16 + state);
        ++n_eval; n_eval_grad += want_grad ? 1 : 0;
        eval_point<NT, SC, P, AXIS, LIN, MINW>(kp, cx, ev, ew, c, icm, ya, yb, want_grad, state == ST_OUTER, o PROF_PASS);
        bool step_begin = false;

        if (state == ST_INIT0) {
            // cost + gradient at u; perturbation for the local Lipschitz estimate
            cost = P::uni(o.psi); gv = o.gv; gw = o.gw;
            double h0, h1;
            panoc_lip_perturbation<P>(cx, vl, uv, uw, h0, h1, nh);
            ev = uv + h0; ew = uw + h1; want_grad = true; state = ST_INIT1;
            continue;
        } else if (state == ST_INIT1) {
            panoc_lip_estimate<P>(cx, o.gv - gv, o.gw - gw, nh, Lip, gamma, ig, sigma);
            panoc_envelope_sums<P>(kp, vl, uv, uw, gamma, gv, gw, hv, hw, gg, d2h);
            step_begin = true;
        }
#if !MPC_STEP_LOOP
        else if (state == ST_LIP) {
            const double cost_half = o.psi;
            if (panoc_lip_test_fails(cx, cost_half, cost, ip, ig, nfpr) && lip_it < MAX_LIP_IT && Lip < KC(K_MAX_LIP)) {
                lb.flush();  // invalidate the L-BFGS buffer
                panoc_lip_update<P>(kp, vl, uv, uw, gv, gw, Lip, gamma, ig, hv, hw, rv_, rw_, d2h, nfpr, ip);
                ++lip_it;
                ev = hv; ew = hw; want_grad = false;
                continue;
            }
            sigma = P::uni(KC(K_SIGMA) * ig);
#ifdef MPC_TRACE
            tr_psi_u = cost;
#endif
            // ---- L-BFGS buffer update with (state = u, g = gamma*fpr)
            double lb_pr = 0.0;  // products of the rows of [S; Y] with gamma*fpr (Gram form): from update() to direction()
            lb.template update<P, NT, MEMT>(cx, kp, vl, lane, uv, uw, rv_, rw_, nfpr, lm, lb_pr);
            wave_sync();
            if (iter == 0) {
                // first iteration: no line search, u <- u_half
                uv = hv; uw = hw;
                ev = uv; ew = uw; want_grad = true; state = ST_NOLS;
                continue;
            }
            lb.template direction<P, NT, MEMT, LBG>(cx, kp, vl, lane, rv_, rw_, lm, lb_pr, dv, dw);
            // ---- line search on the forward-backward envelope
            rhs = P::uni(panoc_fbe_rhs(cost, gamma, ig, gg, d2h, sigma, nfpr));
            tau = 1.0; nls = 0;
            ev = panoc_trial(uv, rv_, dv, tau); ew = panoc_trial(uw, rw_, dw, tau);  // u_plus
            want_grad = true; state = ST_LS;
            continue;
        }
#endif
        else if (state == ST_NOLS) {
            cost = P::uni(o.psi); gv = o.gv; gw = o.gw;
            panoc_envelope_sums<P>(kp, vl, uv, uw, gamma, gv, gw, hv, hw, gg, d2h);
#ifdef MPC_TRACE
            tr_write(-1, 1.0);
#endif
            ++iter;
            step_begin = true;
        }
#if !MPC_STEP_LOOP
        else if (state == ST_LS) {
            // (ev, ew) is the trial point u_plus
            cost = P::uni(o.psi); gv = o.gv; gw = o.gw;
            panoc_envelope_sums<P>(kp, vl, ev, ew, gamma, gv, gw, hv, hw, gg, d2h);
            const double lhs = panoc_fbe(cost, gamma, ig, gg, d2h);
            if (lhs > rhs && nls < MAX_LS_IT) {
                tau = P::uni(tau * 0.5); ++nls;
                ev = panoc_trial(uv, rv_, dv, tau); ew = panoc_trial(uw, rw_, dw, tau);
                want_grad = true;
                continue;
            }
            // MAX_LS_IT halvings without acceptance.  ls_fallback = 0: the last trial point (tau = 2^-10) is the next
            // iterate (what the published code effectively does); ls_fallback = 1: tau = 0, the
            // point u - gamma*fpr is evaluated and taken unconditionally.
            if (kp.ls_fallback == 1 && lhs > rhs && tau != 0.0) {
                tau = 0.0;
                ev = uv - rv_; ew = uw - rw_;
                want_grad = true;
                continue;
            }
            uv = ev; uw = ew;
#ifdef MPC_TRACE
            tr_write(nls, tau);
#endif
            ++iter;
            step_begin = true;
        }
#endif
        else {  // ST_OUTER: evaluated at the inner solution (c, y still those of the inner problem)
            inner_total += num_iter;
            last_fpr = nfpr;
            f_final = P::uni(o.f);
            double ypa, ypb;
            alm_multiplier_step<P>(kp, vl, o.F1a, o.F1b, ya, yb, c, ypa, ypb, dy_norm_plus);
            f2_norm_plus = P::uni(sqrt(o.nrm2F2));
            bool done = false;
            // converged: status = status of the last inner problem; the outer-iteration cap overrides it
            if (alm_exit(cx, kp, alm_iteration, dy_norm_plus, f2_norm_plus, akkt_tol, c) || num_outer == kp.max_outer) {
                if (num_outer == kp.max_outer) status = 1;
                done = true;
            } else if (kp.max_ticks > 0 && wall_clock64() - t_start > kp.max_ticks) {
                status = 2;
                done = true;
            }
            if (done) {
                if (vl && io.y_out) {
                    io.y_out[(size_t)b * 2 * N + lane] = ypa;
                    io.y_out[(size_t)b * 2 * N + N + lane] = ypb;
                }
                break;
            }
            if (!alm_stalled(cx, kp, alm_iteration, dy_norm_plus, dy_norm, f2_norm_plus, f2_norm)) {
                c = P::uni(c * kp.penalty_update); icm = P::uni(1.0 / fmax(c, 1.0));
            }
            akkt_tol = P::uni(fmax(akkt_tol * kp.tol_update, kp.tol));
            set_inner_tolerance(cx, lane, akkt_tol);
            ++alm_iteration; ++num_outer;
            dy_norm = dy_norm_plus; f2_norm = f2_norm_plus;
            ya = clampd(ypa, -KC(K_YBOUND), KC(K_YBOUND)); yb = clampd(ypb, -KC(K_YBOUND), KC(K_YBOUND));  // y <- Proj_Y(y+)
            // reset the PANOC cache for the next inner problem
            lb.flush(); tau = 1.0; Lip = 0; sigma = 0; gamma = 0; ig = 0; iter = 0;
            num_iter = 0; cont_iters = true; cont_time = true;
            ev = uv; ew = uw; want_grad = true; state = ST_INIT0;
            continue;
        }

#if MPC_STEP_LOOP
        if (step_begin) {
            // ---- the PANOC steps of an inner problem: a loop of its own with its own call sites of eval_point (Lipschitz test,
            //      line search); the state machine above is left with the evaluations that happen ten times per solve.
            count_step = state != ST_INIT1;   // a full step has completed: the solver loop's bookkeeping
            bool to_nols = false;
            for (;;) {
                if (CAN_YIELD && MPC_YIELD_STEP && kp.yield_from > 0 && (num_iter & kp.yield_mask) == 0) {
                    // the launch is draining when all but yield_cap of its problems have finished: hand this one to the latency kernel
                    if (__hip_atomic_load(io.counts + CNT_FINISHED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= kp.yield_from) {
                        if (lane == 0) yslot = __hip_atomic_fetch_add(io.counts + CNT_YIELDED, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        yslot = __builtin_amdgcn_readfirstlane(yslot);
                        if (yslot < kp.yield_cap) { yielded = true; break; }   // (the list cannot overflow: at most yield_cap problems are unfinished)
                    }
                }
                bool inner_done = false;
                if (count_step) {
                    if (cont_iters && cont_time) {
                        ++num_iter;
                        cont_iters = num_iter < kp.max_inner;
                        if (kp.max_ticks > 0) cont_time = (wall_clock64() - t_start) <= kp.max_ticks;
                    } else {
                        inner_done = true;
                    }
                }
                count_step = true;
                if (!inner_done && panoc_step_residual<P>(cx, kp, vl, uv, uw, hv, hw, gv, gw, gamma, iter, akkt_tol, rv_, rw_, nfpr, ip))
                    inner_done = true;
                if (inner_done) break;
                lip_it = 0;
                for (;;) {   // Lipschitz test at the half step
                    PROF_MARK(10 + ST_LIP);
                    PROF_COUNT(16 + ST_LIP);
                    ++n_eval;
                    eval_point<NT, SC, P, AXIS, LIN, MINW>(kp, cx, hv, hw, c, icm, ya, yb, false, false, o PROF_PASS);
                    const double cost_half = o.psi;
                    if (panoc_lip_test_fails(cx, cost_half, cost, ip, ig, nfpr) && lip_it < MAX_LIP_IT && Lip < KC(K_MAX_LIP)) {
                        lb.flush();  // invalidate the L-BFGS buffer
                        panoc_lip_update<P>(kp, vl, uv, uw, gv, gw, Lip, gamma, ig, hv, hw, rv_, rw_, d2h, nfpr, ip);
                        ++lip_it;
                        continue;
                    }
                    break;
                }
                sigma = P::uni(KC(K_SIGMA) * ig);
#ifdef MPC_TRACE
                tr_psi_u = cost;
#endif
                double lb_pr = 0.0;
                lb.template update<P, NT, MEMT>(cx, kp, vl, lane, uv, uw, rv_, rw_, nfpr, lm, lb_pr);
                wave_sync();
                if (iter == 0) { to_nols = true; break; }   // first iteration of an inner problem: no line search (state machine)
                lb.template direction<P, NT, MEMT, LBG>(cx, kp, vl, lane, rv_, rw_, lm, lb_pr, dv, dw);
                rhs = P::uni(panoc_fbe_rhs(cost, gamma, ig, gg, d2h, sigma, nfpr));
                tau = 1.0; nls = 0;
                ev = panoc_trial(uv, rv_, dv, tau); ew = panoc_trial(uw, rw_, dw, tau);  // u_plus
                for (;;) {   // line search on the forward-backward envelope
                    PROF_MARK(10 + ST_LS);
                    PROF_COUNT(16 + ST_LS);
                    ++n_eval; ++n_eval_grad;
                    eval_point<NT, SC, P, AXIS, LIN, MINW>(kp, cx, ev, ew, c, icm, ya, yb, true, false, o PROF_PASS);
                    cost = P::uni(o.psi); gv = o.gv; gw = o.gw;
                    panoc_envelope_sums<P>(kp, vl, ev, ew, gamma, gv, gw, hv, hw, gg, d2h);
                    const double lhs = panoc_fbe(cost, gamma, ig, gg, d2h);
                    if (lhs > rhs && nls < MAX_LS_IT) {
                        tau = P::uni(tau * 0.5); ++nls;
                        ev = panoc_trial(uv, rv_, dv, tau); ew = panoc_trial(uw, rw_, dw, tau);
                        continue;
                    }
                    if (kp.ls_fallback == 1 && lhs > rhs && tau != 0.0) {   // see ST_LS
                        tau = 0.0;
                        ev = uv - rv_; ew = uw - rw_;
                        continue;
                    }
                    break;
                }
                uv = ev; uw = ew;
#ifdef MPC_TRACE
                tr_write(nls, tau);
#endif
                ++iter;
            }
            if (CAN_YIELD && MPC_YIELD_STEP && yielded) break;
            if (to_nols) {
                uv = hv; uw = hw;
                ev = uv; ew = uw; want_grad = true; state = ST_NOLS;
                continue;
            }
            // inner problem finished: the feasible half step is the result
            status = !cont_iters ? 1 : (!cont_time ? 2 : 0);
            uv = hv; uw = hw;
            ev = uv; ew = uw; want_grad = false; state = ST_OUTER;
        }
    }
    if (CAN_YIELD && yielded) {
        // ---- the state of the iteration at this step boundary -> the problem's workspace record; the continuation launch
        //      (solve_kernel_team with io.ylist set) reads it back and goes on with the bookkeeping of the step loop above
        double* yr = io.ws + (size_t)b * kp.ws_stride + kp.ws_yield;
        if (state == ST_INIT0) {   // start of an inner problem: point, multipliers and the scalars of the outer loop
            if (vl) {
                double2* v = reinterpret_cast<double2*>(yr + YS_SCALARS + lane * YS_VECW);
                v[0] = make_double2(uv, uw); v[3] = make_double2(ya, yb);
            }
            if (lane == 0) {
                yr[YS_C] = c; yr[YS_AKKT] = akkt_tol; yr[YS_DYN] = dy_norm; yr[YS_F2N] = f2_norm;
                yr[YS_ELAPSED] = (double)(wall_clock64() - t_start);
                yr[YS_FLAGS] = 16;
                yr[YS_ALMIT] = alm_iteration; yr[YS_NOUTER] = num_outer; yr[YS_INNERTOT] = inner_total;
                yr[YS_NEVAL] = n_eval; yr[YS_NEVALG] = n_eval_grad;
                yr[YS_LBHEAD] = lb.head;   // the ring keeps its position across a flush, and the Gram form sums the rows in slot order
#ifdef MPC_TRACE
                yr[YS_TRN] = tr_n;
#endif
            }
            // The record first, then the list entry: the continuation may run CONCURRENTLY (another stream, any XCD) and takes the
            // entry as soon as it sees it.  One release fence per wavefront covers the stores of all its lanes.
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            if (lane == 0) {
                __hip_atomic_store(io.ylist + yslot, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(io.counts + CNT_LISTED, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
#ifdef MPC_PROFILE
            prof.mark(22); prof.flush();
#endif
            return;
        }
#if MPC_YIELD_STEP
        if (vl) {
            double2* v = reinterpret_cast<double2*>(yr + YS_SCALARS + lane * YS_VECW);
            v[0] = make_double2(uv, uw); v[1] = make_double2(gv, gw); v[2] = make_double2(hv, hw); v[3] = make_double2(ya, yb);
        }
        if (lane == 0) {
            yr[YS_C] = c; yr[YS_GAMMA] = gamma; yr[YS_IG] = ig; yr[YS_LIP] = Lip; yr[YS_COST] = cost;
            yr[YS_GG] = gg; yr[YS_D2H] = d2h; yr[YS_AKKT] = akkt_tol; yr[YS_NFPR] = nfpr;
            yr[YS_DYN] = dy_norm; yr[YS_F2N] = f2_norm;
            yr[YS_HGAMMA] = lb.hgamma; yr[YS_ELAPSED] = (double)(wall_clock64() - t_start);
            yr[YS_ITER] = iter; yr[YS_NUMITER] = num_iter;
            yr[YS_FLAGS] = (cont_iters ? 1 : 0) | (cont_time ? 2 : 0) | (count_step ? 4 : 0) | (lb.first ? 8 : 0);
            yr[YS_ALMIT] = alm_iteration; yr[YS_NOUTER] = num_outer; yr[YS_INNERTOT] = inner_total;
            yr[YS_NEVAL] = n_eval; yr[YS_NEVALG] = n_eval_grad; yr[YS_LBACTIVE] = lb.active; yr[YS_LBHEAD] = lb.head;
#ifdef MPC_TRACE
            yr[YS_TRN] = tr_n; yr[YS_TRPSI] = tr_psi_u;
#endif
        }
        {   // the L-BFGS scalars that live in LDS: rho and the Gram matrices (or nothing worth keeping, two-loop form)
            double* yl = yr + YS_SCALARS + N * YS_VECW;
            const int nrho = yield_even_c(mem), ngg = gg_doubles_c(N, mem, gram_shape(N, mem));
            for (int i = lane; i < nrho; i += P::W) yl[i] = lm.LRHO[i];
            for (int i = lane; i < ngg; i += P::W) yl[nrho + i] = lm.GG[i];
        }
        if (!LBG) {   // L-BFGS-in-LDS build: the ring and the previous (u, gamma fpr) travel through the record as well
            double* wr = io.ws + (size_t)b * kp.ws_stride;
            for (int i = lane; i < (2 * mem + 1) * N * 2; i += P::W) wr[kp.ws_lbs + i] = lm.LM[i];
            for (int i = lane; i < N * 4; i += P::W) wr[kp.ws_lold + i] = lm.LOLD[i];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // record, L-BFGS scalars and ring first, then the list entry (see above)
        if (lane == 0) {
            __hip_atomic_store(io.ylist + yslot, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(io.counts + CNT_LISTED, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
#ifdef MPC_PROFILE
        prof.mark(22); prof.flush();
#endif
#endif
        return;
    }
#else
        if (step_begin) {
            bool inner_done = false;
            if (state != ST_INIT1) {  // a full step has completed: the solver loop's bookkeeping
                if (cont_iters && cont_time) {
                    ++num_iter;
                    cont_iters = num_iter < kp.max_inner;
                    if (kp.max_ticks > 0) cont_time = (wall_clock64() - t_start) <= kp.max_ticks;
                } else {
                    inner_done = true;
                }
            }
            if (!inner_done) {
                if (panoc_step_residual<P>(cx, kp, vl, uv, uw, hv, hw, gv, gw, gamma, iter, akkt_tol, rv_, rw_, nfpr, ip)) {
                    inner_done = true;
                } else {
                    lip_it = 0;
                    ev = hv; ew = hw; want_grad = false; state = ST_LIP;
                    continue;
                }
            }
            // inner problem finished: the feasible half step is the result
            status = !cont_iters ? 1 : (!cont_time ? 2 : 0);
            uv = hv; uw = hw;
            ev = uv; ew = uw; want_grad = false; state = ST_OUTER;
        }
    }

#endif
#ifdef MPC_PROFILE
    prof.mark(22); prof.flush();
#endif
    // a NaN / inf anywhere in the iteration ends here too (every comparison with it is false): report it the way
    // OpEn does (SolverError::NotFiniteComputation) instead of returning the garbage as a solution
    if (P::any(vl && !(isfinite(uv) && isfinite(uw))) || !isfinite(f_final)) status = 3;
    // ---- write results (coalesced per problem)
    if (vl) {
        io.u[(size_t)b * 2 * N + 2 * lane] = uv;
        io.u[(size_t)b * 2 * N + 2 * lane + 1] = uw;
    }
    if (lane == 0) {
        io.cost[b] = f_final;
        io.status[b] = status;
        if (io.inner_it) io.inner_it[b] = inner_total;
        if (io.evals) { io.evals[2 * b] = n_eval; io.evals[2 * b + 1] = n_eval_grad; }
        if (io.outer_it) io.outer_it[b] = num_outer;
        if (io.fpr) io.fpr[b] = last_fpr;
        if (io.f2norm) io.f2norm[b] = f2_norm_plus;
        if (io.ms) io.ms[b] = (double)(wall_clock64() - t_start) * 1e-5;  // 100 MHz ticks -> ms
        if (kp.yield_from > 0) __hip_atomic_fetch_add(io.counts + CNT_FINISHED, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int NT, bool SC, bool LBG, int MINW, bool AXIS = false, bool LIN = false>
__global__ __launch_bounds__(WAVE, MINW) void solve_kernel_pair(KParams kp, BatchPtrs io, int B) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    solve_body<NT, SC, LBG, Solo<NT>, AXIS, LIN, MINW>(kp, io, B, lds);
}
// two problems per wavefront (grid = ceil(B / 2)); 2 wavefronts per SIMD = the same 16 resident problems per CU
template <int NT, bool SC, bool LBG>
__global__ __launch_bounds__(WAVE, 2) void solve_kernel_duo(KParams kp, BatchPtrs io, int B) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    solve_body<NT, SC, LBG, Duo<NT>>(kp, io, B, lds);
}


}  // namespace mpcgpu
