// mpc_team.hpp -- latency path of the batched NMPC solver: ONE problem per WORKGROUP of four wavefronts.
//
// The reference calls its solver for one robot at a time (src/interface_mpc.py:82-88, one `solver.run(p)` per tick); a
// fleet of a few robots is a batch of a few problems.  In that regime the GPU is empty and what counts is how long ONE
// solve takes.  A PANOC iteration is a chain of ~4 dependent psi evaluations (Lipschitz test at the half step, then line
// search trials tau = 1, 1/2, 1/4, ...; mpc_kernels.hpp solve_body) -- the chain, not the arithmetic, is the latency.
// This kernel runs the SAME iteration with the evaluations of one step side by side:
//   * the four wavefronts are replicas of one state machine (same scalars, same vectors, bit for bit);
//   * at the start of a step the L-BFGS update and the two-loop direction are computed SPECULATIVELY (they do not depend on
//     the Lipschitz test; if the test fails the buffer is flushed anyway), then in ONE pass wavefront 0 evaluates psi at the
//     half step while wavefronts 1..3 evaluate the trial points tau = 1, 1/2, 1/4;
//   * the first accepted trial in sequential order wins; its wavefront publishes point, gradient and half step through LDS
//     and every replica adopts them.  No trial accepted: the next four trials, again in one pass (tau down to 2^-10);
//   * a failed Lipschitz test discards the speculation and the replicas continue with the sequential rule.
// Every number is computed by the same device functions (eval_point, the step helpers below) on the same inputs as in the
// one-wavefront kernel, so the results are BITWISE those of solve_kernel_pair (tests/test_gpu_latency.py); the library is
// built with -ffp-contract=on so that floating-point contraction does not depend on the code around an expression.
//
// Also fused here: the parameter compaction (prep_problem) -- one launch per call -- and the LDS carve comes from the
// configured maxima (general dynamic-obstacle tables), so the host never reads a count back before the launch.
#pragma once
#include "mpc_kernels.hpp"

namespace mpcgpu {

constexpr int TEAM_WAVES = 4;   // wavefronts per problem of the latency kernel proper; the mid-batch form runs 2 (template parameter TW)
constexpr int TEAM_XCH = 8;  // doubles per wavefront in the exchange area

// 2^-k, k = 0 .. 1022, built from its bit pattern: the line-search step lengths tau = 1, 1/2, 1/4, ... of the throughput kernel are
// exact halvings; a library exp2 is not REQUIRED to return the exact power (and need not be the same code in every kernel)
__device__ __forceinline__ double pow2_neg(int k) { return __hiloint2double((1023 - k) << 20, 0); }

enum { TS_INIT0 = 0, TS_INIT1, TS_FIRST, TS_SPEC, TS_LIPSEQ, TS_BATCH, TS_FALLBACK, TS_OUTER };

// TW = 4: wavefront 0 takes the Lipschitz test, wavefronts 1-3 the trials tau = 1, 1/2, 1/4 (1.38 passes per PANOC step on the
// benchmark scenes).  TW = 2 (round 3, batches between 2 and 4 problems per compute unit): Lipschitz test + tau = 1 side by side
// -- 67 % of the steps end there -- then two trials per pass; a workgroup is half as big, so twice as many problems are resident
// and such a batch runs in ONE round instead of two.  Same device functions, same bits.
// RESUME (round 5, tail promotion -- see YIELD in mpc_kernels.hpp): the same kernel is the continuation of a throughput launch
// when it is handed the list of promoted problems (io.ylist != NULL; a RUN-TIME switch on purpose: the loop below is then the very
// machine code every latency-kernel test exercises, not a second instantiation of it).  Workgroup g takes problem ylist[g]
// (workgroups beyond the device-side list length leave at once), reads the state its wavefront of the throughput kernel left in
// the workspace record at the start of an inner problem -- point, multipliers, penalty, tolerance, the outer loop's scalars and
// counters, the ring position -- into every replica and starts that inner problem.  Same step functions, same state: bitwise
// what the throughput kernel would have gone on to write.
// Wavefronts per SIMD the latency kernel is compiled for: 2 = as many registers as it likes (192 / 210 VGPRs, no scratch), 3 = 168
// VGPRs (a few values of the outer loop in scratch) and three four-wavefront teams per compute unit instead of two.
#ifndef MPC_TEAM_WPE
#define MPC_TEAM_WPE 2
#endif
#if MPC_TEAM_WPE == 2
#define MPC_TEAM_ATTR
#else
#define MPC_TEAM_ATTR __attribute__((amdgpu_waves_per_eu(MPC_TEAM_WPE, MPC_TEAM_WPE)))
#endif
// Round 6: the PANOC steps of the common path -- residual, speculative L-BFGS step, the pass with the Lipschitz test and the first
// trials, further trial passes -- in a loop of their OWN with their own call sites of the evaluation (what the throughput kernel got in
// round 3, MPC_STEP_LOOP); the rare states (first step of an inner problem, Lipschitz updates, the tau = 0 fallback, outer steps) stay
// in the state machine around it.  Same step functions on the same values in the same order: same bits.  0 = one loop around one call
// site (rounds 3-5).
#ifndef MPC_TEAM_STEP_LOOP
#define MPC_TEAM_STEP_LOOP 1
#endif
// CONCURRENT (round 5): with kp.yield_persist the kernel runs on a second stream WHILE the throughput launch drains (behind
// tail_gate_kernel, which holds it back until the launch starts to promote).  Workgroup g waits for list entry g -- entries appear
// in index order and workgroups are dispatched in index order, so the resident ones wait for the entries that come next -- or until
// every problem of the throughput launch has finished or is listed (FINISHED + LISTED = yield_total: the list is final and entry g
// is not part of it), solves the problem and marks the entry done (-2).  A sweep launch of the same kernel (yield_persist = 0)
// behind both takes whatever is still >= 0 in the list: normally nothing.  No wait is unbounded (wall-clock limits): a scheduling
// surprise costs time, not a hang.
template <int NT, int TW>
__global__ __launch_bounds__(WAVE * TW) MPC_TEAM_ATTR void solve_kernel_team(KParams kp, BatchPtrs io, int B) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    using P = Solo<NT>;
    constexpr bool SC = false;  // general tables: no batch-wide shape information is needed before the launch
    const int N = NT ? NT : kp.N;  // compile-time horizon (0 = runtime horizon from KParams)
    const bool resume = io.ylist != nullptr;
    const bool persist = resume && kp.yield_persist != 0;
    int b = blockIdx.x;
    if (persist) {
        double* claim = lds + kp.l_xch + 3;   // a spare double of wavefront 0's exchange block
        if (threadIdx.x == 0) {
            const int idx = blockIdx.x;
            int bb = -1;
            const long long t0 = wall_clock64();
            for (;;) {
                int v = __hip_atomic_load(io.ylist + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v >= 0) { bb = v; break; }
                const int fin = __hip_atomic_load(io.counts + CNT_FINISHED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int listed = __hip_atomic_load(io.counts + CNT_LISTED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (fin + listed >= kp.yield_total) {   // the list is final: is this entry part of it?
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // (here, not in every look: an acquire invalidates this CU's vector cache)
                    v = __hip_atomic_load(io.ylist + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (v >= 0) bb = v;
                    break;
                }
                if (wall_clock64() - t0 > 50000000LL) {        // 0.5 s (100 MHz): give up, the sweep takes the entry
                    atomicAdd(io.counts + CNT_TIMEOUTS, 1);
                    break;
                }
                __builtin_amdgcn_s_sleep(64);
            }
            claim[0] = (double)bb;
        }
        __syncthreads();
        b = (int)uniform(claim[0]);
        if (b < 0) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // the record the throughput kernel wrote before the entry
    } else if (resume) {
        int n = io.counts[CNT_YIELDED];
        n = n < kp.yield_cap ? n : kp.yield_cap;
        if (b >= n) return;
        b = __builtin_amdgcn_readfirstlane(io.ylist[b]);
        if (b < 0) return;   // solved by the concurrent continuation (-2) -- this launch is the sweep behind it
    }
    if (b >= B) return;
    long long t_start = wall_clock64();
    // The wavefront's index is wave-uniform, and the compiler is told so (v_readfirstlane): every `wid` branch below -- whose
    // evaluation point, who publishes, who hands over -- is then a scalar branch instead of a divergent region that a wavefront
    // walks through with an empty EXEC mask.
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & (WAVE - 1), mem = kp.mem;
    double* ws = io.ws + (size_t)b * kp.ws_stride;
    if (kp.reserved) {   // tables sized from promised / measured maxima: a problem beyond them is reported, not solved (solve_body)
        const bool over = (int)uniform(ws[H_KS]) > kp.mKs || (int)uniform(ws[H_KF]) > kp.mKf || (int)uniform(ws[H_KD]) > kp.mKd;
        if (over) {
            const double nan = __builtin_nan("");
            if (wid == 0 && lane < N) {
                io.u[(size_t)b * 2 * N + 2 * lane] = 0.0; io.u[(size_t)b * 2 * N + 2 * lane + 1] = 0.0;
                if (io.y_out) { io.y_out[(size_t)b * 2 * N + lane] = nan; io.y_out[(size_t)b * 2 * N + N + lane] = nan; }
            }
            if (threadIdx.x == 0) {
                io.cost[b] = nan; io.status[b] = 4;
                if (io.inner_it) io.inner_it[b] = 0;
                if (io.outer_it) io.outer_it[b] = 0;
                if (io.evals) { io.evals[2 * b] = 0; io.evals[2 * b + 1] = 0; }
                if (io.fpr) io.fpr[b] = nan;
                if (io.f2norm) io.f2norm[b] = nan;
                if (io.ms) io.ms[b] = 0.0;
            }
            return;
        }
    }
    // io.p == NULL: the record was already written by the tracker's assembly kernel (mpc_tracker.hpp)
    if (wid == 0 && io.p) prep_problem(kp, ParamVector{io.p + (size_t)b * kp.np}, ws, io.counts, lane);
    __syncthreads();

    // ---- problem context: tables shared by the four wavefronts, work areas and L-BFGS memory per wavefront
    Ctx cx;
    double* mine = lds + wid * kp.l_wstride;
    {
        cx.lane = lane; cx.vl = lane < N; cx.il = true; cx.ik = lane % N; cx.isub = lane / N;
        double* hd = lds + kp.l_hd;
        if (threadIdx.x < KC_BASE) hd[threadIdx.x] = ws[threadIdx.x];
        if (threadIdx.x >= WAVE && threadIdx.x < WAVE + 27) hd[KC_BASE + threadIdx.x - WAVE] = KTAB[threadIdx.x - WAVE];
        cx.hd = hd;
        cx.akkt = lds + kp.l_xch + wid * TEAM_XCH + 2;   // this wavefront's own slot of the exchange area ([0], [1]: verdict banks, [7]: clock)
        auto U = [&](int i) { return uniform(ws[i]); };
        cx.Ks = (int)U(H_KS); cx.Kf = (int)U(H_KF); cx.Kd = (int)U(H_KD);
        cx.pad_f = U(H_NPF) > 0.0; cx.pad_d = U(H_NPD) > 0.0;
        cx.terminal = U(H_QN) != 0.0 || U(H_QTHN) != 0.0;
        cx.vref = cx.vl ? ws[kp.ws_vref + lane] : 0.0;
        cx.seg = lds + kp.l_seg; cx.stc = lds + kp.l_stc; cx.fxy = lds + kp.l_fxy;
        cx.dyn = lds + kp.l_dyn; cx.dync = cx.dyn;
        cx.pos = mine + kp.l_pos; cx.H = mine + kp.l_H; cx.W = mine + kp.l_W; cx.part = mine + kp.l_part; cx.stash = mine + kp.l_stash;
        cx.bal = mine + kp.l_bal;
        set_balanced_trips(cx, N, WAVE);
        const int T = WAVE * TW;
        for (int i = threadIdx.x; i < N * SEGW; i += T) cx.seg[i] = ws[kp.ws_seg + i];
        for (int i = threadIdx.x; i < cx.Ks * STCW; i += T) cx.stc[i] = ws[kp.ws_stc + i];
        for (int i = threadIdx.x; i < cx.Kf * N * 2; i += T) cx.fxy[i] = ws[kp.ws_fxy + i];
        for (int i = threadIdx.x; i < cx.Kd * N * DYNW; i += T) cx.dyn[i] = ws[kp.ws_dyn + i];
    }
    __syncthreads();
    constexpr int MEMT = MemOf<NT>::value;
    LbMem lm;
    lm.LM = mine + kp.l_S;  // [S; Y] contiguous
    lm.LRHO = mine + kp.l_rho;
    lm.LALPHA = mine + kp.l_alpha;
    lm.LOLD = mine + kp.l_old;
    lm.GG = mine + kp.l_gg;
    lm.XA = cx.pos;
    for (int i = lane; i < (2 * mem + 1) * N; i += WAVE) reinterpret_cast<double2*>(lm.LM)[i] = make_double2(0.0, 0.0);
    double* XF = lds + kp.l_xch;                          // [TEAM_WAVES][TEAM_XCH]: per-wavefront verdicts of a pass
    double* XV = XF + TW * TEAM_XCH;                      // [N][6] + 4: the winner's point, gradient, half step and scalars
    const bool vl = cx.vl;
    const int MAX_LIP_IT = 10, MAX_LS_IT = 10;

    double uv = 0.0, uw = 0.0, ya = 0.0, yb = 0.0;
    if (vl) {
        if (io.u0) { uv = io.u0[(size_t)b * 2 * N + 2 * lane]; uw = io.u0[(size_t)b * 2 * N + 2 * lane + 1]; }
        if (io.y0) { ya = io.y0[(size_t)b * 2 * N + lane]; yb = io.y0[(size_t)b * 2 * N + N + lane]; }
    }
    double c = kp.init_penalty;
    if (io.c0) { const double c0 = io.c0[b]; if (c0 > 0.0) c = c0; }
    c = uniform(c);
    double icm = uniform(1.0 / fmax(c, 1.0));
    ya = clampd(ya, -KC(K_YBOUND), KC(K_YBOUND)); yb = clampd(yb, -KC(K_YBOUND), KC(K_YBOUND));
    double gv = 0, gw = 0, hv = 0, hw = 0, rv_ = 0, rw_ = 0, dv = 0, dw = 0;
    double gamma = 0, ig = 0, Lip = 0, sigma = 0, cost = 0, nfpr = 0, rhs = 0, nh = 1, gg = 0, d2h = 0, ip = 0;
    double akkt_tol = kp.init_tol;
    int iter = 0, num_iter = 0, lip_it = 0, nls = 0, t0 = 0, head_spec = 0;
    bool cont_iters = true, cont_time = true;
    typename LbfgsOf<false, NT>::type lb;
    int alm_iteration = 0, num_outer = 1, inner_total = 0, status = 0;
    double dy_norm = 0, dy_norm_plus = 0, f2_norm = 0, f2_norm_plus = 0, last_fpr = 0, f_final = 0;
    int n_eval = 0, n_eval_grad = 0;  // evaluations of the SEQUENTIAL algorithm (the speculative ones are not counted)

#ifdef MPC_TRACE
    // decision trace (trace builds, tests): the record of solve_body, written by wavefront 0 when a PANOC step completes
    int tr_n = 0;
    double tr_psi_u = 0.0;
    auto tr_write = [&](int nls_, double tau_) {
        if (wid == 0 && lane == 0 && io.trace && tr_n < io.trace_cap) {
            double* r = io.trace + ((size_t)b * io.trace_cap + tr_n) * TRACE_W;
            r[0] = alm_iteration; r[1] = iter; r[2] = c; r[3] = Lip; r[4] = gamma; r[5] = nfpr; r[6] = tr_psi_u;
            r[7] = lip_it; r[8] = lb.active; r[9] = nls_; r[10] = tau_; r[11] = cost;
        }
        ++tr_n;
    };
#define TEAM_TRACE(nls_, tau_) tr_write(nls_, tau_)
#define TEAM_TRACE_PSI() tr_psi_u = cost
#else
#define TEAM_TRACE(nls_, tau_)
#define TEAM_TRACE_PSI()
#endif
    int state = TS_INIT0;
    bool resume_pending = false;
    if (resume) {
        // the boundary the throughput kernel left at (solve_body, `yielded`): every replica reads the same record
        const double* yr = ws + kp.ws_yield;
        auto YU = [&](int i) { return uniform(yr[i]); };
        c = YU(YS_C); icm = uniform(1.0 / fmax(c, 1.0));
        const int fl = (int)YU(YS_FLAGS);
        akkt_tol = YU(YS_AKKT); dy_norm = YU(YS_DYN); f2_norm = YU(YS_F2N);
        t_start -= (long long)YU(YS_ELAPSED);
        alm_iteration = (int)YU(YS_ALMIT); num_outer = (int)YU(YS_NOUTER); inner_total = (int)YU(YS_INNERTOT);
        n_eval = (int)YU(YS_NEVAL); n_eval_grad = (int)YU(YS_NEVALG);
        lb.head = (int)YU(YS_LBHEAD);   // the ring keeps its position across a flush, and the Gram form sums the rows in slot order
#ifdef MPC_TRACE
        tr_n = (int)YU(YS_TRN);
#endif
        if (fl & 16) {
            // left at the start of an inner problem: point and multipliers; the PANOC cache and the L-BFGS buffer start empty
            if (vl) {
                const double2* v = reinterpret_cast<const double2*>(yr + YS_SCALARS + lane * YS_VECW);
                const double2 a = v[0], y = v[3];
                uv = a.x; uw = a.y; ya = y.x; yb = y.y;
            }
        }
#if MPC_YIELD_STEP
        else {
        gamma = YU(YS_GAMMA); ig = YU(YS_IG); Lip = YU(YS_LIP); cost = YU(YS_COST); gg = YU(YS_GG); d2h = YU(YS_D2H);
        nfpr = YU(YS_NFPR); lb.hgamma = YU(YS_HGAMMA);
        iter = (int)YU(YS_ITER); num_iter = (int)YU(YS_NUMITER);
        cont_iters = (fl & 1) != 0; cont_time = (fl & 2) != 0; lb.first = (fl & 8) != 0;
        state = (fl & 4) ? TS_BATCH : TS_INIT1;   // TS_INIT1 = no step has completed yet in this inner problem (the bookkeeping is skipped)
        lb.active = (int)YU(YS_LBACTIVE);
#ifdef MPC_TRACE
        tr_psi_u = YU(YS_TRPSI);
#endif
        if (vl) {
            const double2* v = reinterpret_cast<const double2*>(yr + YS_SCALARS + lane * YS_VECW);
            const double2 a = v[0], g = v[1], hh = v[2], y = v[3];
            uv = a.x; uw = a.y; gv = g.x; gw = g.y; hv = hh.x; hw = hh.y; ya = y.x; yb = y.y;
        }
        // the L-BFGS memory of this replica: ring + zero row, previous (u, gamma fpr), rho, Gram matrices
        for (int i = lane; i < (2 * mem + 1) * N; i += WAVE)
            reinterpret_cast<double2*>(lm.LM)[i] = reinterpret_cast<const double2*>(ws + kp.ws_lbs)[i];
        for (int i = lane; i < N * 4; i += WAVE) lm.LOLD[i] = ws[kp.ws_lold + i];
        const double* yl = yr + YS_SCALARS + N * YS_VECW;
        const int nrho = yield_even_c(mem), ngg = gg_doubles_c(N, mem, gram_shape(N, mem));
        for (int i = lane; i < nrho; i += WAVE) lm.LRHO[i] = yl[i];
        for (int i = lane; i < ngg; i += WAVE) lm.GG[i] = yl[nrho + i];
        resume_pending = true;
        }
#endif
        wave_sync();
    }
    set_inner_tolerance(cx, lane, akkt_tol);
    double ev = uv, ew = uw;
    bool want_grad = true;
    EvalOut o;
#ifdef MPC_PROFILE
    Prof prof; prof.start();  // the phase profiler is a tool of the throughput kernel; here it only keeps the signature
#endif

    // verdicts of a pass: every wavefront publishes one flag, all read the four of them
    // Two alternating flag banks (like the wall-clock slots below): with ONE barrier per call, a fast wavefront may already
    // write its flag of pass k + 1 while a slow one still reads the four flags of pass k.
    int pub_slot = 0;
    auto publish = [&](double flag, double* all) {
        PROF_MARK(15);
        if (lane == 0) XF[wid * TEAM_XCH + pub_slot] = flag;
        __syncthreads();
        for (int j = 0; j < TW; ++j) all[j] = uniform(XF[j * TEAM_XCH + pub_slot]);
        pub_slot ^= 1;
        PROF_MARK(10);
    };
    // the winning wavefront hands over (point, gradient, half step; cost, ||grad||^2, ||gradient step - half step||^2)
    auto adopt = [&](int winner, double tcost, double tgg, double td2h, double thv, double thw) {
        PROF_MARK(13);
        if (wid == winner) {
            if (vl) {
                double* r = XV + lane * 6;
                r[0] = ev; r[1] = ew; r[2] = o.gv; r[3] = o.gw; r[4] = thv; r[5] = thw;
            }
            if (lane == 0) { XV[N * 6] = tcost; XV[N * 6 + 1] = tgg; XV[N * 6 + 2] = td2h; }
        }
        __syncthreads();
        if (vl) {
            const double* r = XV + lane * 6;
            uv = r[0]; uw = r[1]; gv = r[2]; gw = r[3]; hv = r[4]; hw = r[5];
        } else {
            uv = 0.0; uw = 0.0; gv = 0.0; gw = 0.0; hv = 0.0; hw = 0.0;
        }
        cost = uniform(XV[N * 6]); gg = uniform(XV[N * 6 + 1]); d2h = uniform(XV[N * 6 + 2]);
        PROF_MARK(11);
    };
    // The wall clock differs between wavefronts by a few ticks: wavefront 0 decides for all (two alternating slots, so that a
    // wavefront that is one barrier behind still reads the verdict it was meant to read).
    int clock_slot = 0;
    auto time_left = [&]() -> bool {
        double* slot = XF + (1 + clock_slot) * TEAM_XCH - 1;
        clock_slot ^= 1;
        if (lane == 0 && wid == 0) *slot = (wall_clock64() - t_start) <= kp.max_ticks ? 1.0 : 0.0;
        __syncthreads();
        return uniform(*slot) != 0.0;
    };
    auto trial_point = [&](double t) { ev = panoc_trial(uv, rv_, dv, t); ew = panoc_trial(uw, rw_, dw, t); };

    // The verdict of a pass, shared by the state machine and the step loop (MPC_TEAM_STEP_LOOP).  After the pass that carries the Lipschitz
    // test and the first trials: 0 = the test failed (state = TS_LIPSEQ, the half step is the next point), 1 = a trial was adopted (the next
    // step begins), 2 = none accepted (state = TS_BATCH, the next trials are set).
    auto spec_verdict = [&]() -> int {
            // wavefront 0: psi at the half step (Lipschitz test); wavefronts 1..3: trials tau = 1, 1/2, 1/4
            double thv = 0.0, thw = 0.0, tgg = 0.0, td2h = 0.0, flag;
            if (wid == 0) {
                flag = (panoc_lip_test_fails(cx, o.psi, cost, ip, ig, nfpr) && lip_it < MAX_LIP_IT && Lip < KC(K_MAX_LIP)) ? 1.0 : 0.0;
            } else {
                panoc_envelope_sums<P>(kp, vl, ev, ew, gamma, o.gv, o.gw, thv, thw, tgg, td2h);
                flag = panoc_fbe(uniform(o.psi), gamma, ig, tgg, td2h) > rhs ? 0.0 : 1.0;  // 1 = accepted
            }
            double all[TW];
            publish(flag, all);
            ++n_eval;
            if (all[0] != 0.0) {
                // Lipschitz test failed: the speculative pair and direction are void (the buffer is flushed).  The ring position
                // goes back as well: the Gram form sums the rows in slot order, so the bits depend on where the ring stands, and
                // the one-wavefront kernel never stored this pair.
                lb.flush();
                lb.head = head_spec;
                panoc_lip_update<P>(kp, vl, uv, uw, gv, gw, Lip, gamma, ig, hv, hw, rv_, rw_, d2h, nfpr, ip);
                ++lip_it;
                ev = hv; ew = hw; want_grad = false; state = TS_LIPSEQ;
                return 0;
            }
            int winner = 0;
            for (int j = TW - 1; j >= 1; --j) if (all[j] != 0.0) winner = j;
            if (winner) {
                nls = winner - 1;
                n_eval += nls + 1; n_eval_grad += nls + 1;
                TEAM_TRACE_PSI();
                adopt(winner, uniform(o.psi), tgg, td2h, thv, thw);
                TEAM_TRACE(nls, pow2_neg(nls));
                ++iter;
                return 1;
            }
            n_eval += TW - 1; n_eval_grad += TW - 1;
            TEAM_TRACE_PSI();
            t0 = TW - 1;
            trial_point(pow2_neg(t0 + wid));
            want_grad = true; state = TS_BATCH;
            return 2;
    };
    // ... after a pass of trials only: 1 = adopted, 2 = the next trials are set, 3 = ten halvings without acceptance under the tau = 0 reading
    // (state = TS_FALLBACK, the point u - gamma fpr is set).
    auto batch_verdict = [&]() -> int {
            // trials t0 + wid, tau = 2^-t; t = MAX_LS_IT is taken unconditionally (ls_fallback = 0) or leads to the tau = 0 point
            double thv, thw, tgg, td2h;
            panoc_envelope_sums<P>(kp, vl, ev, ew, gamma, o.gv, o.gw, thv, thw, tgg, td2h);
            const int t = t0 + wid;
            const bool accepted = !(panoc_fbe(uniform(o.psi), gamma, ig, tgg, td2h) > rhs);
            const double flag = t > MAX_LS_IT ? 0.0 : (accepted ? 1.0 : (t == MAX_LS_IT ? 2.0 : 0.0));  // 2 = last trial, rejected
            double all[TW];
            publish(flag, all);
            int winner = -1;
            for (int j = TW - 1; j >= 0; --j) if (all[j] != 0.0) winner = j;
            if (winner >= 0) {
                nls = t0 + winner;
                n_eval += winner + 1; n_eval_grad += winner + 1;
                if (all[winner] == 2.0 && kp.ls_fallback == 1) {
                    TEAM_TRACE_PSI();
                    // 10 halvings without acceptance, tau = 0 reading: u - gamma*fpr is evaluated and taken
                    ev = uv - rv_; ew = uw - rw_;
                    want_grad = true; state = TS_FALLBACK;
                    return 3;
                }
                TEAM_TRACE_PSI();
                adopt(winner, uniform(o.psi), tgg, td2h, thv, thw);
                TEAM_TRACE(nls, pow2_neg(nls));
                ++iter;
                return 1;
            }
            n_eval += TW; n_eval_grad += TW;
            t0 += TW;
            trial_point(pow2_neg(t0 + wid));
            return 2;
    };

    for (;;) {
        bool step_begin = false;
        if (MPC_YIELD_STEP && resume_pending) {
            resume_pending = false;
            step_begin = true;
        } else {
        PROF_MARK(19);    // (profile builds: slots 10 verdict barrier, 11 adoption, 12 L-BFGS pair + direction, 14 step residual, 15 end of a pass -> its
        PROF_COUNT(16);   //  verdict, 18 bookkeeping of a completed step, 19 direction -> next pass incl. the loop's back edge, 13 the rest; counts: 16 passes,
                          //  17 PANOC steps; wavefront 0 of every team reports)
        eval_point<NT, SC, P>(kp, cx, ev, ew, c, icm, ya, yb, want_grad, state == TS_OUTER, o PROF_PASS);

        if (state == TS_INIT0) {
            ++n_eval; ++n_eval_grad;
            cost = uniform(o.psi); gv = o.gv; gw = o.gw;
            double h0, h1;
            panoc_lip_perturbation<P>(cx, vl, uv, uw, h0, h1, nh);
            ev = uv + h0; ew = uw + h1; want_grad = true; state = TS_INIT1;
            continue;
        } else if (state == TS_INIT1) {
            ++n_eval; ++n_eval_grad;
            panoc_lip_estimate<P>(cx, o.gv - gv, o.gw - gw, nh, Lip, gamma, ig, sigma);
            panoc_envelope_sums<P>(kp, vl, uv, uw, gamma, gv, gw, hv, hw, gg, d2h);
            step_begin = true;
        } else if (state == TS_FIRST || state == TS_LIPSEQ) {
            // psi at the half step, the same point on every wavefront.  TS_FIRST (first step of an inner problem): evaluated
            // WITH the gradient, because once the Lipschitz test passes this very point becomes the iterate.
            ++n_eval;
            const double cost_half = o.psi;
            if (panoc_lip_test_fails(cx, cost_half, cost, ip, ig, nfpr) && lip_it < MAX_LIP_IT && Lip < KC(K_MAX_LIP)) {
                lb.flush();
                panoc_lip_update<P>(kp, vl, uv, uw, gv, gw, Lip, gamma, ig, hv, hw, rv_, rw_, d2h, nfpr, ip);
                ++lip_it;
                ev = hv; ew = hw;  // want_grad stays as it is
                continue;
            }
            sigma = uniform(KC(K_SIGMA) * ig);
            TEAM_TRACE_PSI();
            double lb_pr = 0.0;
            lb.template update<P, NT, MEMT>(cx, kp, vl, lane, uv, uw, rv_, rw_, nfpr, lm, lb_pr);
            wave_sync();
            if (state == TS_FIRST) {
                // no line search on the first step: u <- u_half, whose psi and gradient were just evaluated
                ++n_eval; ++n_eval_grad;
                uv = hv; uw = hw;
                cost = uniform(o.psi); gv = o.gv; gw = o.gw;
                panoc_envelope_sums<P>(kp, vl, uv, uw, gamma, gv, gw, hv, hw, gg, d2h);
                TEAM_TRACE(-1, 1.0);
                ++iter;
                step_begin = true;
            } else {
                lb.template direction<P, NT, MEMT, true>(cx, kp, vl, lane, rv_, rw_, lm, lb_pr, dv, dw);
                rhs = panoc_fbe_rhs(cost, gamma, ig, gg, d2h, sigma, nfpr);
                t0 = 0;
                trial_point(pow2_neg(t0 + wid));
                want_grad = true; state = TS_BATCH;
                continue;
            }
        } else if (state == TS_SPEC) {
            if (spec_verdict() == 1) step_begin = true; else continue;
        } else if (state == TS_BATCH) {
            if (batch_verdict() == 1) step_begin = true; else continue;
        } else if (state == TS_FALLBACK) {
            ++n_eval; ++n_eval_grad;
            uv = ev; uw = ew;
            cost = uniform(o.psi); gv = o.gv; gw = o.gw;
            panoc_envelope_sums<P>(kp, vl, uv, uw, gamma, gv, gw, hv, hw, gg, d2h);
            TEAM_TRACE(nls, 0.0);
            ++iter;
            step_begin = true;
        } else {  // TS_OUTER
            ++n_eval;
            inner_total += num_iter;
            last_fpr = nfpr;
            f_final = uniform(o.f);
            double ypa, ypb;
            alm_multiplier_step<P>(kp, vl, o.F1a, o.F1b, ya, yb, c, ypa, ypb, dy_norm_plus);
            f2_norm_plus = uniform(sqrt(o.nrm2F2));
            bool done = false;
            if (alm_exit(cx, kp, alm_iteration, dy_norm_plus, f2_norm_plus, akkt_tol, c) || num_outer == kp.max_outer) {
                if (num_outer == kp.max_outer) status = 1;
                done = true;
            } else if (kp.max_ticks > 0 && !time_left()) {
                status = 2;
                done = true;
            }
            if (done) {
                if (wid == 0 && vl && io.y_out) {
                    io.y_out[(size_t)b * 2 * N + lane] = ypa;
                    io.y_out[(size_t)b * 2 * N + N + lane] = ypb;
                }
                break;
            }
            if (!alm_stalled(cx, kp, alm_iteration, dy_norm_plus, dy_norm, f2_norm_plus, f2_norm)) {
                c = uniform(c * kp.penalty_update); icm = uniform(1.0 / fmax(c, 1.0));
            }
            akkt_tol = uniform(fmax(akkt_tol * kp.tol_update, kp.tol));
            set_inner_tolerance(cx, lane, akkt_tol);
            ++alm_iteration; ++num_outer;
            dy_norm = dy_norm_plus; f2_norm = f2_norm_plus;
            ya = clampd(ypa, -KC(K_YBOUND), KC(K_YBOUND)); yb = clampd(ypb, -KC(K_YBOUND), KC(K_YBOUND));
            lb.flush(); Lip = 0; sigma = 0; gamma = 0; ig = 0; iter = 0;
            num_iter = 0; cont_iters = true; cont_time = true;
            ev = uv; ew = uw; want_grad = true; state = TS_INIT0;
            continue;
        }
        }   // (evaluation + its state; skipped once when the kernel resumes at a step boundary)

        if (step_begin) {
            bool to_machine = false;   // leave for the state machine and its call site of the evaluation
            for (;;) {                 // MPC_TEAM_STEP_LOOP: one trip per PANOC step while the common path holds (else: a single trip)
                bool inner_done = false;
                if (state != TS_INIT1) {
                    if (cont_iters && cont_time) {
                        ++num_iter;
                        cont_iters = num_iter < kp.max_inner;
                        // the wall clock is looked at every 16th step (a barrier per look: wavefront 0 decides for all); the throughput
                        // kernel looks every step -- a time-out is not reproducible to the step in either
                        if (kp.max_ticks > 0 && (num_iter & 15) == 0) cont_time = time_left();
                    } else {
                        inner_done = true;
                    }
                }
                if (inner_done) break;
                PROF_MARK(18);
                PROF_COUNT(17);
                const bool solved = panoc_step_residual<P>(cx, kp, vl, uv, uw, hv, hw, gv, gw, gamma, iter, akkt_tol, rv_, rw_, nfpr, ip);
                PROF_MARK(14);
                if (solved) break;
                lip_it = 0;
                if (iter == 0) {
                    ev = hv; ew = hw; want_grad = true; state = TS_FIRST;
                    to_machine = true;
                    break;
                }
                // speculation: pair update and direction before the Lipschitz test is known
                sigma = uniform(KC(K_SIGMA) * ig);
                head_spec = lb.head;
                double lb_pr = 0.0;
                PROF_MARK(13);
                lb.template update<P, NT, MEMT>(cx, kp, vl, lane, uv, uw, rv_, rw_, nfpr, lm, lb_pr);
                wave_sync();
                lb.template direction<P, NT, MEMT, true>(cx, kp, vl, lane, rv_, rw_, lm, lb_pr, dv, dw);
                PROF_MARK(12);
                rhs = panoc_fbe_rhs(cost, gamma, ig, gg, d2h, sigma, nfpr);
                if (wid == 0) { ev = hv; ew = hw; want_grad = false; }
                else { trial_point(pow2_neg(wid - 1)); want_grad = true; }
                state = TS_SPEC;
#if MPC_TEAM_STEP_LOOP
                // the pass with the Lipschitz test and the first trials, then passes of trials only: call sites of their own
                PROF_MARK(19);
                PROF_COUNT(16);
                eval_point<NT, SC, P>(kp, cx, ev, ew, c, icm, ya, yb, want_grad, false, o PROF_PASS);
                int act = spec_verdict();
                if (act == 2) {
                    for (;;) {
                        PROF_MARK(19);
                        PROF_COUNT(16);
                        eval_point<NT, SC, P>(kp, cx, ev, ew, c, icm, ya, yb, want_grad, false, o PROF_PASS);
                        act = batch_verdict();
                        if (act != 2) break;
                    }
                }
                if (act == 1) continue;   // a trial was adopted: the next step begins
#endif
                to_machine = true;        // (step loop: a Lipschitz update or the tau = 0 fallback -- rare; else: every pass is the state machine's)
                break;
            }
            if (to_machine) continue;
            status = !cont_iters ? 1 : (!cont_time ? 2 : 0);
            uv = hv; uw = hw;
            ev = uv; ew = uw; want_grad = false; state = TS_OUTER;
        }
    }

#ifdef MPC_PROFILE
    if (wid == 0) { prof.mark(13); prof.flush(); }
#endif
    if (__ballot(vl && !(isfinite(uv) && isfinite(uw))) != 0ull || !isfinite(f_final)) status = 3;
    if (wid == 0) {
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
            if (io.ms) io.ms[b] = (double)(wall_clock64() - t_start) * 1e-5;
            if (persist) __hip_atomic_store(io.ylist + blockIdx.x, -2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // done: nothing for the sweep
        }
    }
}

// Holds the stream of the concurrent continuation back until the throughput launch starts to promote (FINISHED >= yield_from) AND
// every problem of that launch has begun (STARTED = total).  Workgroups of the latency kernel that arrive earlier would take
// registers and LDS from a GPU that is still full; and a launch of short problems is limited by the dispatch rate, not by residency:
// its last workgroups are still waiting when it starts to promote, and teams that fill the compute units must not wait for list
// entries those very workgroups would write (seen once before this condition existed: a tick of 2.2 s = the wall-clock limit of
// the wait in solve_kernel_team at the time).  One lane, asleep between looks; bounded by a wall-clock limit like every wait here.
__global__ __launch_bounds__(WAVE) void tail_gate_kernel(int* counts, int yield_from, int total, long long max_ticks, int gradual) {
    if (threadIdx.x != 0) return;
    const long long t0 = wall_clock64();
    for (;;) {
        const int fin = __hip_atomic_load(counts + CNT_FINISHED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool promoting = fin >= yield_from ||
                               (gradual && __hip_atomic_load(counts + CNT_YIELDED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 0);   // (gradual promotion: the first entry opens the gate)
        if (promoting && __hip_atomic_load(counts + CNT_STARTED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= total) break;
        if (fin + __hip_atomic_load(counts + CNT_LISTED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= total) break;   // the launch is over
        if (wall_clock64() - t0 >= max_ticks) { atomicAdd(counts + CNT_TIMEOUTS, 1); break; }
        __builtin_amdgcn_s_sleep(127);
    }
}

}  // namespace mpcgpu
