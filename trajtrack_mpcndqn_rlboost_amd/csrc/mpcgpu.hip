// mpcgpu.hip -- C-ABI (include/mpcgpu.h) over the gfx950 kernels in mpc_kernels.hpp.
//
// Host side only: argument validation, device buffers, workspace + LDS layout, launches, timing events.
// No CPU fallback: every entry point needs a HIP device.
#include "../../include/mpcgpu.h"
#include "mpc_kernels.hpp"
#include "mpc_team.hpp"
#include "mpc_tracker.hpp"
#include "mpc_order.hpp"

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <unordered_map>
#include <mutex>
#include <vector>

using namespace mpcgpu;

namespace {

thread_local std::string g_create_error;

struct DevBuf {
    void* ptr = nullptr;
    size_t cap = 0;
};

#ifndef MPC_TAIL_GRADUAL_DEFAULT
#define MPC_TAIL_GRADUAL_DEFAULT 0
#endif

struct Handle {
    mpcgpu_config cfg{};
    KParams kp{};
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // prep start/end, solve start/end, [4] between the throughput kernel and its continuation
    bool tail_timed = false;   // ev[4] was recorded by the last solve (tail promotion enqueued a continuation launch)
    bool timing_valid = false;
    std::string err;
    // grow-only device buffers
    DevBuf ws, counts, evals, perm, bins, nlist, ylist, trace, p, u0, y0, c0, u, cost, status, inner, outer, fpr, f2, y, ms, xi, psi, f, grad, F1, F2;
    int order = 1;      // MPCGPU_OPT_ORDER: 0 problems are dispatched as given, 1 longest first by the previous call's evaluation counts
    int evals_B = 0;    // batch size of the call whose evaluation counts `evals` holds (0: none)
    int last_ordered = 0;  // the last throughput launch used a permutation
    int* h_counts = nullptr;  // pinned
    int last_shape[4] = {0, 0, 0, 0};
    int last_B = 0;  // batch size of the last solve (for mpcgpu_last_eval_counts)
    bool shape_const = true;  // of the batch prepared last
    bool axis_aligned = false;  // ... and every active dynamic row of it has angle 0 (known only after a count read-back)
    bool linear = false;        // ... and moves on a straight line (linear centre tables, mpc_kernels.hpp prep_problem)
    int opt_linear = 1;         // MPCGPU_OPT_LINEAR_TABLES: 1 = use the linear centre tables where they pay (N_hor = 40, large batches)
    bool last_linear = false;   // the last launch ran a kernel with linear centre tables
    int last_min_waves = 0;   // launch-bounds variant of the last solve (3 or 4 wavefronts per SIMD)
    int num_cus = 256;
    int team_max_batch = -1;  // MPCGPU_OPT_TEAM_BATCH: largest batch solved by the latency kernel (-1: 4 x number of CUs)
    int last_team = 0;        // wavefronts per problem of the latency kernel the last solve ran (4 or 2); 0: throughput kernel
    int pairing = -1;   // MPCGPU_OPT_PAIRING: -1 automatic, 0 one problem per wavefront, 1 two per wavefront (N_hor = 20)
    int last_pairing = 0;  // layout of the last solve / cost_grad launch
    int yield_opt = -1;  // MPCGPU_OPT_TAIL_PROMOTION: -1 automatic (twice the resident four-wavefront teams), 0 off, > 0 that many problems
    int yield_poll = 16; // ... builds with -DMPC_YIELD_STEP=1 only: the finished-counter is also polled every this many PANOC steps (power of two)
    int yield_waves = 0; // MPCGPU_OPT_TAIL_WAVES: wavefronts per promoted problem (0: four; two for an explicit capacity beyond four times the residency)
    int last_yield_cap = 0;  // capacity of the continuation launch of the last solve (0: none was enqueued)
    int tail_gradual = MPC_TAIL_GRADUAL_DEFAULT;   // MPCGPU_OPT_TAIL_GRADUAL: finished problems per gradually promoted one (0 = off)
    int tail_concurrent = 1; // MPCGPU_OPT_TAIL_CONCURRENT: 1 = the continuation runs on `side` while the throughput launch drains (mpc_team.hpp CONCURRENT)
    hipStream_t side = nullptr;                      // stream of the concurrent continuation
    hipEvent_t ev_fork = nullptr, ev_join = nullptr; // launch stream -> side (records written), side -> launch stream (promoted problems solved)
    bool last_concurrent = false;
    bool in_call = false;    // inside solve_common right now (guarded by g_handles_mu, see another_launch_in_flight)
    int trace_cap = 0;  // -DMPC_TRACE builds: PANOC steps recorded per problem (0 = tracing off)
    // mpcgpu_reserve_shape: upper bounds of active rows promised by the caller -> no count read-back before the launch
    bool capturing = false;  // the launch stream of the current call is being captured into a hipGraph: no event records
    bool reserved = false;
    int res_shape[4] = {0, 0, 0, 0};  // max static, fleet, dynamic rows; 0 = shape-constant rows, 1 = they may change shape, 2 = shape-constant AND axis-aligned
    hipStream_t last_stream = nullptr;  // launch stream of the last solve: compared only, never dereferenced (it may be gone by now)
    bool last_captured = false;         // the last solve was recorded into a hipGraph (no completion event exists for it)
    std::unordered_map<const void*, int> lds_attr;  // kernel -> largest dynamic-LDS size opted into (hipFuncSetAttribute once, not per launch)
};

// Handles of this process (mpcgpu_create / mpcgpu_destroy): a solve call looks at the others to see whether another launch is in
// flight on its device (the concurrent continuation is for ONE launch at a time, see solve_common).  "In flight" = the other
// handle is INSIDE a solve call right now (`in_call`, set and cleared under the mutex: two threads that enter their calls together
// both see the other one and both keep the continuation behind their launch -- never both beside it), or its end-of-call event
// has not completed yet.  Other PROCESSES on the same GPU and graph replays of another handle are invisible here: such users set
// MPCGPU_OPT_TAIL_CONCURRENT = 0 (include/mpcgpu.h); the cost of not doing so is time (a starved side stream), never results.
std::mutex g_handles_mu;
std::vector<Handle*> g_handles;

struct InCall {   // marks the handle as inside solve_common for the lifetime of the object
    Handle* h;
    explicit InCall(Handle* h_) : h(h_) { std::lock_guard<std::mutex> lock(g_handles_mu); h->in_call = true; }
    ~InCall() { std::lock_guard<std::mutex> lock(g_handles_mu); h->in_call = false; }
    InCall(const InCall&) = delete;
    InCall& operator=(const InCall&) = delete;
};

bool another_launch_in_flight(const Handle* h) {
    std::lock_guard<std::mutex> lock(g_handles_mu);
    for (const Handle* o : g_handles) {
        if (o == h || o->device != h->device) continue;
        if (o->in_call) return true;
        if (o->ev[3] && hipEventQuery(o->ev[3]) == hipErrorNotReady) {
            (void)hipGetLastError();   // "not ready" is an answer, not an error to carry along
            return true;
        }
    }
    return false;
}

int fail(Handle* h, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (h) h->err = buf; else g_create_error = buf;
    return code;
}

#define HIP_OK(h, call)                                                                                 \
    do {                                                                                                \
        hipError_t e_ = (call);                                                                         \
        if (e_ != hipSuccess) return fail(h, -10, "%s failed: %s", #call, hipGetErrorString(e_));       \
    } while (0)

// Grow-only buffer.  A buffer that must grow may still be read by work enqueued earlier on ANY stream the caller used
// with this handle, so the device is drained before the old allocation is released (hipFree would do so implicitly; it is
// spelled out because correctness depends on it).  Growth is geometric, so steady-state calls never get here.
int ensure(Handle* h, DevBuf& b, size_t bytes) {
    if (bytes <= b.cap) return 0;
    if (h->capturing)
        return fail(h, -6, "a device buffer must grow to %zu bytes while the stream is being captured: call mpcgpu_reserve_batch "
                           "(or run one eager call of at least this batch size) before the capture", bytes);
    if (b.ptr) {
        HIP_OK(h, hipDeviceSynchronize());
        HIP_OK(h, hipFree(b.ptr));
    }
    b.ptr = nullptr;
    const size_t want = bytes > b.cap + b.cap / 2 ? bytes : b.cap + b.cap / 2;
    b.cap = 0;
    HIP_OK(h, hipMalloc(&b.ptr, want));
    b.cap = want;
    return 0;
}

inline hipStream_t pick_stream(Handle* h, void* stream) {
    return stream == MPCGPU_STREAM_OWN ? h->stream : (hipStream_t)stream;
}

inline int even(int x) { return (x + 1) & ~1; }

// dynamic LDS beyond 64 KiB must be opted into per kernel: once per (handle, kernel, size), never inside a captured region
int opt_in_lds(Handle* h, const void* kern, size_t bytes) {
    if (bytes <= 64 * 1024) return 0;
    auto it = h->lds_attr.find(kern);
    if (it != h->lds_attr.end() && it->second >= (int)bytes) return 0;
    if (h->capturing)
        return fail(h, -6, "this launch needs %zu bytes of dynamic LDS, which has to be opted into outside a stream capture: run one "
                           "eager call with the same configuration first", bytes);
    HIP_OK(h, hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    h->lds_attr[kern] = (int)bytes;
    return 0;
}
inline bool stream_is_capturing(hipStream_t s) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(s, &cap);
    return cap != hipStreamCaptureStatusNone;
}

// Two problems per wavefront (Duo layout, mpc_kernels.hpp) exist for the compiled horizon N_hor = 20.  Measured on the
// benchmark batch it is SLOWER than one problem per wavefront (1631 vs 1274 ms for 32 768 solves: 11 % fewer VALU
// instructions per solve, but 219 VGPRs and two LDS carves leave 2 wavefronts per SIMD and the kernel turns latency bound;
// DESIGN.md section 7), so the automatic rule keeps one problem per wavefront and the layout stays an explicit option.
// The choice never depends on the batch: results must not change with the batch a problem travels in.
// The kernels with a compile-time horizon (N_hor = 20, 40: the reference's yaml files) also fix the L-BFGS memory at the
// reference's 10 (mpc_kernels.hpp MemOf); every other configuration runs the generic kernel.
inline int compiled_horizon(const Handle* h) { return (h->kp.N == 20 || h->kp.N == 40) && h->kp.mem == 10 ? h->kp.N : 0; }
inline bool duo_available(const Handle* h) { return compiled_horizon(h) == 20; }
inline bool use_duo(const Handle* h) { return duo_available(h) && h->pairing == 1; }

// Where the L-BFGS memory (2 x 10 x 2N doubles per problem) lives.  true: in the problem's workspace record (HBM,
// L2-resident while the solve runs), which frees 6.4 KB of LDS per wavefront at N = 20; false: in LDS.
#ifndef MPC_LBFGS_IN_WORKSPACE
#define MPC_LBFGS_IN_WORKSPACE 1
#endif
constexpr bool LBFGS_IN_WORKSPACE = MPC_LBFGS_IN_WORKSPACE != 0;
#ifndef MPC_TRY_FOUR_WAVES
#define MPC_TRY_FOUR_WAVES 1
#endif

#ifndef MPC_FOUR_WAVES_FROM
#define MPC_FOUR_WAVES_FROM (4 * MPC_MIN_WAVES)   // problems per compute unit from which the 128-VGPR build is taken (more than the other build holds at once)
#endif

void fill_static_params(Handle* h) {
    const mpcgpu_config& c = h->cfg;
    KParams& k = h->kp;
    const int N = c.N;
    k.N = N;
    k.Nother = c.Nother; k.Nstcobs = c.Nstcobs; k.Ndynobs = c.Ndynobs; k.mem = c.lbfgs_mem;
    k.max_inner = c.max_inner; k.max_outer = c.max_outer;
    k.ls_fallback = 0;
    k.stall_rule = 0;
    k.ts = c.ts; k.inv_ts = 1.0 / c.ts;
    k.vmin = c.lin_vel_min; k.vmax = c.lin_vel_max; k.wmax = c.ang_vel_max;
    k.amin = c.lin_acc_min; k.amax = c.lin_acc_max; k.aamax = c.ang_acc_max;
    k.W2 = c.vehicle_width * c.vehicle_width; k.social = c.social_margin; k.fleetw = c.fleet_weight;
    k.tol = c.tol; k.delta_tol = c.delta_tol; k.init_tol = c.init_tol; k.init_penalty = c.init_penalty;
    k.penalty_update = c.penalty_update; k.tol_update = c.tol_update; k.suff_decrease = c.suff_decrease;
    k.max_ticks = c.max_duration_us > 0.0 ? (long long)(c.max_duration_us * 100.0) : 0;
    // parameter layout: mpc_generator.py:179-188
    k.r0 = 18;
    k.c0 = k.r0 + 4 * N;
    k.os0 = k.c0 + 3 * N * c.Nother;
    k.od0 = k.os0 + c.Nstcobs * c.nstcobs;
    k.qs0 = k.od0 + c.Ndynobs * c.ndynobs * N;
    k.qd0 = k.qs0 + N;
    k.np = k.qd0 + N;
    // workspace layout (worst-case strides)
    int o = HDR;
    k.ws_vref = o; o += even(N);
    k.ws_seg = o; o += even(N * SEGW);
    k.ws_stc = o; o += c.Nstcobs * STCW;
    k.ws_fxy = o; o += c.Nother * N * 2;
    k.ws_dyn = o; o += even(c.Ndynobs * N * DYNW);
    k.ws_qd = o; o += even(N);
    k.ws_alpha = o; o += even(c.Ndynobs);
    k.ws_dynl = o; o += c.Ndynobs * DYNL;             // linear centre tables: row constants ...
    k.ws_dynr = o; o += even((c.Ndynobs * N + 1) / 2);   // ... and one 32-bit residual word per (row, step)
    k.ws_lbs = o; o += c.lbfgs_mem * N * 2;   // L-BFGS memory when it is kept in the workspace (see LBFGS_IN_WORKSPACE)
    k.ws_lby = o; o += c.lbfgs_mem * N * 2;
    o += N * 2;                               // one row of zeros behind [S; Y] (Gram form: padded row of pass 2)
    k.ws_lold = o; o += N * 4;
    // tail promotion (mpc_kernels.hpp YIELD): the iteration state a problem leaves behind when it moves to the latency kernel
    k.ws_yield = o; o += YS_SCALARS + N * YS_VECW + even(c.lbfgs_mem) + gg_doubles_c(N, c.lbfgs_mem, gram_shape(N, c.lbfgs_mem));
    k.ws_stride = (o + 15) & ~15;
    k.yield_from = 0; k.yield_cap = 0; k.yield_mask = 15; k.yield_persist = 0; k.yield_total = 0; k.yield_grad = 0;
}

// sizes of the fixed regions: mpc_kernels.hpp (part_doubles_c, stash_doubles_c, fixed_lds) -- shared with the kernels
int part_doubles(const KParams& k) { return part_doubles_c(k.N, k.mem); }
bool gram_layout(const KParams& k) { return gram_shape(k.N, k.mem); }
int stash_doubles(const KParams& k, int minw = 3) { return stash_doubles_c(k.N, k.mem, minw); }

// LDS carve for the batch maxima (doubles; every offset even => 16-byte aligned).
// shape_const: every active dynamic row of the batch keeps (rx, ry, angle, alpha) over the horizon -> 2 doubles per
// (row, step) + 6 per row instead of 9 per (row, step).
// linear: straight-line rows -> 4 doubles per row + one 32-bit word per (row, step) instead of the centres.
// minw: wavefronts per SIMD the kernel is compiled for (the stash of the long horizon depends on it: stash_stride_c).
void fill_lds_layout(KParams& k, int mKs, int mKf, int mKd, bool shape_const, bool lbfgs_in_lds, bool linear = false, int minw = 3) {
    const int N = k.N;
    k.mKs = mKs; k.mKf = mKf; k.mKd = mKd;
    // fixed part first (compile-time offsets in the kernels with a compiled horizon), then the tables that follow the batch
    const FixedLds f = fixed_lds(N, k.mem, lbfgs_in_lds, minw);
    k.l_hd = f.hd; k.l_seg = f.seg; k.l_pos = f.pos; k.l_stash = f.stash;
    k.l_H = f.part; k.l_W = f.W; k.l_part = f.part; k.l_bal = f.bal;
    k.l_rho = f.rho; k.l_alpha = f.gg; k.l_gg = f.gg;
    k.l_S = f.S; k.l_Y = f.Y; k.l_old = f.old;
    int o = f.end;
    k.l_stc = o; o += mKs * STCW;
    k.l_fxy = o; o += mKf * N * 2;
    k.l_dynl = 0;
    if (shape_const && linear) {
        k.l_dyn = o; o += even((mKd * N + 1) / 2);   // residual words
        k.l_dynl = o; o += mKd * DYNL;
        k.l_dync = o; o += even(mKd * DYNC);
        k.l_qd = 0;
    } else if (shape_const) {
        k.l_dyn = o; o += even(mKd * N * DYNP);
        k.l_dync = o; o += even(mKd * DYNC);
        k.l_qd = 0;   // q_dyn lives in the pad double of the segment records (mpc_kernels.hpp load_problem)
    } else {
        k.l_dyn = o; o += even(mKd * N * DYNW);
        k.l_dync = k.l_dyn; k.l_qd = k.l_dyn;
    }
    k.l_total = o;
}

// LDS carve of the latency kernel (mpc_team.hpp): tables for the CONFIGURED maxima (general dynamic-obstacle records), shared
// by the four wavefronts of the workgroup; then the exchange area; then one work block per wavefront (positions, stash, hinge
// matrix / item partials, L-BFGS memory).  Offsets of the work-block fields are those of wavefront 0.
// `tw` wavefronts per problem; tables for (mKs, mKf, mKd) active rows -- the configured maxima (nothing known before the launch) or
// the maxima of the batch / of a reservation (mid-batch form).
void fill_team_layout(KParams& k, int tw, int mKs, int mKf, int mKd) {
    const int N = k.N;
    k.mKs = mKs; k.mKf = mKf; k.mKd = mKd;
    int o = 0;
    k.l_seg = o; o += even(N * SEGW);
    k.l_stc = o; o += k.mKs * STCW;
    k.l_fxy = o; o += k.mKf * N * 2;
    k.l_dyn = o; o += even(k.mKd * N * DYNW);
    k.l_dync = k.l_dyn; k.l_qd = k.l_dyn;
    k.l_hd = o; o += 64;
    k.l_xch = o; o += tw * TEAM_XCH + even(N * 6 + 4);
    const int base = o;
    k.l_pos = o; o += N * 2;
    k.l_stash = o; o += stash_doubles(k);
    const int h_rows = even(k.mKd);
    const int h_sz = h_rows + even(k.mKd), part_sz = part_doubles(k);
    k.l_H = o; k.l_W = o + h_rows; k.l_part = o;
    o += h_sz > part_sz ? h_sz : part_sz;
    k.l_bal = o; o += bal_doubles_c(k.N, k.mem);   // balanced walk of the dynamic rows (eval_point): per wavefront
    k.l_S = o; o += k.mem * N * 2;
    k.l_Y = o; o += k.mem * N * 2 + N * 2;  // + the zero row
    k.l_rho = o; o += even(k.mem);
    k.l_alpha = o;
    k.l_gg = o;
    if (gram_layout(k)) o += even(k.mem * k.mem + k.mem * (k.mem + 1) / 2); else o += even(k.mem);
    k.l_old = o; o += N * 4;
    k.l_wstride = even(o - base);
    k.l_total = base + tw * k.l_wstride;
    k.reserved = 0;
}

// compaction kernel + LDS layout (from the reserved shape, or from a blocking read-back of the batch's active-row
// maxima).  Leaves kp ready for a launch on `s`.
TrackerView tracker_view(const Handle* h, const mpcgpu_tracker* t) {
    TrackerView v{};
    v.B = t->B; v.ref_cap = t->ref_cap; v.action_steps = t->action_steps;
    v.states = t->states; v.goals = t->goals; v.last_actions = t->last_actions; v.ref = t->ref; v.ref_len = t->ref_len;
    v.idx_ref = t->idx_ref; v.stc = t->stc; v.dyn = t->dyn; v.other = t->other; v.pred_states = t->pred_states; v.active = t->active;
    for (int i = 0; i < 10; ++i) v.tuning[i] = t->tuning[i];
    v.base_speed = t->base_speed; v.low_speed = t->low_speed; v.stc_weight = t->stc_weight; v.dyn_weight = t->dyn_weight;
    return v;
}

// `trk` != NULL: the compact records come from the tracker's arrays (tracker_assemble_kernel) instead of parameter vectors
int prepare(Handle* h, int B, const double* d_p, hipStream_t s, BatchPtrs& io, bool allow_reserved,
            const mpcgpu_tracker* trk = nullptr, const double* refs = nullptr) {
    if (int r = ensure(h, h->ws, (size_t)B * h->kp.ws_stride * sizeof(double))) return r;
    if (int r = ensure(h, h->counts, CNT_WORDS * sizeof(int))) return r;
    io.p = d_p;
    io.ws = (double*)h->ws.ptr;
    io.counts = (int*)h->counts.ptr;
    HIP_OK(h, hipMemsetAsync(io.counts, 0, CNT_WORDS * sizeof(int), s));
    if (!h->capturing) HIP_OK(h, hipEventRecord(h->ev[0], s));
    if (trk) hipLaunchKernelGGL(tracker_assemble_kernel, dim3(B), dim3(WAVE), 0, s, h->kp, tracker_view(h, trk), refs, io.ws, io.counts);
    else hipLaunchKernelGGL(prep_kernel, dim3(B), dim3(WAVE), 0, s, h->kp, io, B);
    HIP_OK(h, hipGetLastError());
    if (!h->capturing) HIP_OK(h, hipEventRecord(h->ev[1], s));
    int mKs, mKf, mKd;
    if (allow_reserved && h->reserved) {
        mKs = h->res_shape[0]; mKf = h->res_shape[1]; mKd = h->res_shape[2];
        h->shape_const = h->res_shape[3] != 1;
        h->axis_aligned = h->res_shape[3] == 2;
        h->linear = false;     // not known without a read-back (only the cost_grad test hook asks)
        h->kp.reserved = 1;
    } else {
        if (h->capturing) return fail(h, -6, "stream capture needs mpcgpu_reserve_shape: the automatic LDS carve reads the batch's row counts back");
        HIP_OK(h, hipMemcpyAsync(h->h_counts, io.counts, CNT_WORDS * sizeof(int), hipMemcpyDeviceToHost, s));
        HIP_OK(h, hipStreamSynchronize(s));
        mKs = h->h_counts[CNT_KS]; mKf = h->h_counts[CNT_KF]; mKd = h->h_counts[CNT_KD];
        h->shape_const = h->h_counts[CNT_VARSHAPE] == 0;
        h->axis_aligned = h->shape_const && h->h_counts[CNT_ROTATED] == 0;
        h->linear = h->axis_aligned && h->h_counts[CNT_NONLINEAR] == 0;
        h->kp.reserved = 0;
    }
    fill_lds_layout(h->kp, mKs, mKf, mKd, h->shape_const, !LBFGS_IN_WORKSPACE);
    const int lds_bytes = h->kp.l_total * (int)sizeof(double);
    h->last_shape[0] = mKs; h->last_shape[1] = mKf; h->last_shape[2] = mKd; h->last_shape[3] = lds_bytes;
    if (lds_bytes > 160 * 1024) return fail(h, -5, "LDS carve of %d bytes exceeds 160 KiB", lds_bytes);
    h->last_pairing = use_duo(h) && 2 * lds_bytes <= 160 * 1024 ? 1 : 0;
    return 0;
}

int validate(const mpcgpu_config* c) {
    if (!c) return fail(nullptr, -1, "config is NULL");
    if (c->N < 2 || c->N > 64) return fail(nullptr, -1, "N_hor=%d unsupported (2..64)", c->N);
    if (c->nu != 2 || c->ns != 3) return fail(nullptr, -1, "only the unicycle model (nu=2, ns=3) is supported, got nu=%d ns=%d", c->nu, c->ns);
    if (c->nstcobs != 12) return fail(nullptr, -1, "nstcobs=%d unsupported (12 = 4 edges x (b,a0,a1))", c->nstcobs);
    if (c->ndynobs != 6) return fail(nullptr, -1, "ndynobs=%d unsupported (6)", c->ndynobs);
    if (c->Nother < 0 || c->Nother > 16) return fail(nullptr, -1, "Nother=%d unsupported (0..16)", c->Nother);
    if (c->Nstcobs < 0 || c->Nstcobs > 16) return fail(nullptr, -1, "Nstcobs=%d unsupported (0..16)", c->Nstcobs);
    if (c->Ndynobs < 1 || c->Ndynobs > 32) return fail(nullptr, -1, "Ndynobs=%d unsupported (1..32)", c->Ndynobs);
    if (c->lbfgs_mem < 1 || c->lbfgs_mem > MAX_MEM) return fail(nullptr, -1, "lbfgs_mem=%d unsupported (1..%d)", c->lbfgs_mem, MAX_MEM);
    if (!(c->ts > 0.0)) return fail(nullptr, -1, "ts must be positive");
    if (c->max_inner < 1 || c->max_outer < 1) return fail(nullptr, -1, "max_inner / max_outer must be >= 1");
    if (!(c->tol > 0.0) || !(c->init_tol > 0.0) || !(c->delta_tol > 0.0) || !(c->init_penalty > 0.0))
        return fail(nullptr, -1, "tolerances and initial penalty must be positive");
    if (c->device < 0) return fail(nullptr, -1, "device=%d: a HIP device ordinal is required (no CPU path)", c->device);
    return 0;
}

}  // namespace

extern "C" {

int32_t mpcgpu_abi_version(void) { return MPCGPU_ABI_VERSION; }

int32_t mpcgpu_create(const mpcgpu_config* cfg, void** handle) {
    if (!handle) return fail(nullptr, -1, "handle out-pointer is NULL");
    *handle = nullptr;
    if (int r = validate(cfg)) return r;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, -2, "no HIP device available (%s); libmpcgpu has no CPU fallback",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (cfg->device >= ndev) return fail(nullptr, -2, "device %d requested but only %d present", cfg->device, ndev);
    Handle* h = new (std::nothrow) Handle();
    if (!h) return fail(nullptr, -3, "out of host memory");
    h->cfg = *cfg;
    h->device = cfg->device;
    fill_static_params(h);
#define CREATE_OK(call)                                                                                 \
    do {                                                                                                \
        hipError_t e_ = (call);                                                                         \
        if (e_ != hipSuccess) {                                                                         \
            fail(nullptr, -10, "%s failed: %s", #call, hipGetErrorString(e_));                          \
            delete h;                                                                                   \
            return -10;                                                                                 \
        }                                                                                               \
    } while (0)
    CREATE_OK(hipSetDevice(h->device));
    CREATE_OK(hipDeviceGetAttribute(&h->num_cus, hipDeviceAttributeMultiprocessorCount, h->device));
    CREATE_OK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    for (auto& ev : h->ev) CREATE_OK(hipEventCreate(&ev));
    {   // The side stream of the concurrent continuation must not share a hardware queue with the launch stream (packets of one
        // queue run in order: the gate would sit behind the throughput kernel and the continuation would be the launch behind it
        // again).  The runtime hands out queues per priority class: take the LOWEST one -- workgroups of the latency kernel that
        // wait for a place must never hold back workgroups of the throughput launch (same overlap as the highest class, measured).
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess ||
            hipStreamCreateWithPriority(&h->side, hipStreamNonBlocking, least) != hipSuccess) {
            (void)hipGetLastError();   // no priorities here: an ordinary stream (sharing a queue only costs the overlap, never the result)
            h->side = nullptr;
            CREATE_OK(hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking));
        }
    }
    CREATE_OK(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    CREATE_OK(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
    CREATE_OK(hipHostMalloc((void**)&h->h_counts, CNT_WORDS * sizeof(int), hipHostMallocDefault));
#undef CREATE_OK
    {
        std::lock_guard<std::mutex> lock(g_handles_mu);
        g_handles.push_back(h);
    }
    *handle = h;
    return 0;
}

void mpcgpu_destroy(void* handle) {
    Handle* h = (Handle*)handle;
    if (!h) return;
    {
        std::lock_guard<std::mutex> lock(g_handles_mu);
        for (size_t i = 0; i < g_handles.size(); ++i)
            if (g_handles[i] == h) { g_handles.erase(g_handles.begin() + i); break; }
    }
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->side) (void)hipStreamSynchronize(h->side);
    DevBuf* bufs[] = {&h->ws, &h->counts, &h->evals, &h->perm, &h->bins, &h->nlist, &h->ylist, &h->trace, &h->p, &h->u0, &h->y0, &h->c0, &h->u, &h->cost, &h->status, &h->inner,
                      &h->outer, &h->fpr, &h->f2, &h->y, &h->ms, &h->xi, &h->psi, &h->f, &h->grad, &h->F1, &h->F2};
    for (DevBuf* b : bufs)
        if (b->ptr) (void)hipFree(b->ptr);
    if (h->h_counts) (void)hipHostFree(h->h_counts);
    for (auto& ev : h->ev)
        if (ev) (void)hipEventDestroy(ev);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    if (h->side) (void)hipStreamDestroy(h->side);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

const char* mpcgpu_last_error(void* handle) {
    Handle* h = (Handle*)handle;
    return h ? h->err.c_str() : g_create_error.c_str();
}

int32_t mpcgpu_num_params(void* handle) {
    Handle* h = (Handle*)handle;
    return h ? h->kp.np : -1;
}

}  // extern "C"

namespace {
// One batched solve enqueued on `s`.  The problems come either as parameter vectors `p` (the plugin boundary) or from the
// device-resident tracker state `trk` + the references `refs` its robots track this tick.
int32_t solve_common(Handle* h, int32_t B, const double* p, const mpcgpu_tracker* trk, const double* refs, const double* u0,
                     const double* y0, const double* c0, double* u, double* cost, int32_t* status, int32_t* inner_it,
                     int32_t* outer_it, double* fpr, double* f2norm, double* y_out, double* ms, hipStream_t s) {
    InCall in_call_mark(h);
    h->capturing = stream_is_capturing(s);
    // The evaluation counts the previous call left in `evals` order this launch (MPCGPU_OPT_ORDER) only when that call was
    // enqueued on the SAME stream: another stream gives no ordering between its solve kernel and the kernels that read the counts.
    const bool same_stream_as_last = h->last_stream == s && h->last_B > 0;
    h->last_stream = s;
    h->last_captured = h->capturing;
    BatchPtrs io{};
    // Small batches take the latency kernel: one problem per workgroup of four wavefronts, compaction fused, carve from the
    // configured maxima -- one launch, nothing read back.  Results are bitwise those of the throughput kernel.
    // Default (round 6): up to TWO problems per compute unit -- the four-wavefront form.  The mid-range form (two wavefronts per problem up to four
    // problems per compute unit; MPCGPU_OPT_TEAM_BATCH = 1024 selects it) was the default of rounds 3-5, measured under the "both" reading of the
    // stall rule; under the default reading of this round the throughput kernel with its whole-batch tail promotion is 1-13 % faster from 640 to 1024
    // problems on every family (profiles/r06_team_sweep.txt: a problem whose line search runs all ten halvings costs six passes on two wavefronts)
    const int team_cap = h->team_max_batch >= 0 ? h->team_max_batch : 2 * h->num_cus;
    h->last_team = 0;
    h->last_yield_cap = 0;
    h->tail_timed = false;
    h->last_concurrent = false;
    h->kp.yield_from = 0;
    h->kp.yield_grad = 0;
    bool prepared = false;
#ifdef MPC_TRACE
    if (h->trace_cap > 0) {     // trace builds (tests): one record per PANOC step, from whichever kernel runs
        const size_t tb = (size_t)B * h->trace_cap * TRACE_W * sizeof(double);
        if (int r = ensure(h, h->trace, tb)) return r;
        HIP_OK(h, hipMemsetAsync(h->trace.ptr, 0xFF, tb, s));  // NaN = record not written
        io.trace = (double*)h->trace.ptr;
        io.trace_cap = h->trace_cap;
    }
#endif
#define LAUNCH_TEAM(NT, TW, KT, LDS_T)                                                                               \
    do {                                                                                                             \
        auto kern = solve_kernel_team<NT, TW>;                                                                       \
        if (int r_ = opt_in_lds(h, (const void*)kern, (LDS_T))) return r_;                                           \
        hipLaunchKernelGGL(kern, dim3(B), dim3(WAVE * TW), (LDS_T), s, KT, io, B);                                   \
    } while (0)
#define LAUNCH_TEAM_N(TW, KT, LDS_T)                                                                                 \
    switch (compiled_horizon(h)) {                                                                                   \
        case 20: LAUNCH_TEAM(20, TW, KT, LDS_T); break;                                                              \
        case 40: LAUNCH_TEAM(40, TW, KT, LDS_T); break;                                                              \
        default: LAUNCH_TEAM(0, TW, KT, LDS_T); break;                                                               \
    }
    auto team_done = [&](const KParams& kt, size_t lds_t, int tw) {
        h->last_B = B; h->last_team = tw; h->last_pairing = 0; h->last_min_waves = 1;
        h->evals_B = B; h->last_ordered = 0;
        h->last_shape[0] = kt.mKs; h->last_shape[1] = kt.mKf; h->last_shape[2] = kt.mKd; h->last_shape[3] = (int)lds_t;
    };
    if (B <= team_cap && !use_duo(h)) {
        if (int r = ensure(h, h->ws, (size_t)B * h->kp.ws_stride * sizeof(double))) return r;
        if (int r = ensure(h, h->counts, CNT_WORDS * sizeof(int))) return r;
        if (int r = ensure(h, h->evals, (size_t)B * 2 * sizeof(int32_t))) return r;
        io.p = p; io.ws = (double*)h->ws.ptr; io.counts = nullptr; io.evals = (int32_t*)h->evals.ptr;
        io.u0 = u0; io.y0 = y0; io.c0 = c0; io.u = u; io.cost = cost; io.status = status; io.inner_it = inner_it;
        io.outer_it = outer_it; io.fpr = fpr; io.f2norm = f2norm; io.y_out = y_out; io.ms = ms;
        // (A) up to two problems per compute unit, nothing promised: ONE launch -- four wavefronts per problem, compaction fused,
        //     tables for the configured maxima, nothing read back.
        // Inside a stream capture without a reservation the whole latency range takes this form (the mid-batch form below reads
        // the row counts back, which a capture cannot do): up to four problems per compute unit then run in two rounds.
        if ((B <= 2 * h->num_cus || h->capturing) && !h->reserved) {
            KParams kt = h->kp;
            fill_team_layout(kt, TEAM_WAVES, h->cfg.Nstcobs, h->cfg.Nother, h->cfg.Ndynobs);
            const size_t lds_t = kt.l_total * sizeof(double);
            // prep_problem's static __shared__ tables travel on top of the dynamic carve
            if (lds_t + PREP_STATIC_LDS <= 160 * 1024) {   // else (long horizons with many obstacle slots): the throughput kernel below
                if (!h->capturing) HIP_OK(h, hipEventRecord(h->ev[0], s));
                if (trk) {   // the records come from the tracker's arrays: one more (tiny) launch in front of the solve
                    hipLaunchKernelGGL(tracker_assemble_kernel, dim3(B), dim3(WAVE), 0, s, h->kp, tracker_view(h, trk), refs, io.ws, (int*)nullptr);
                    HIP_OK(h, hipGetLastError());
                }
                if (!h->capturing) { HIP_OK(h, hipEventRecord(h->ev[1], s)); HIP_OK(h, hipEventRecord(h->ev[2], s)); }
                LAUNCH_TEAM_N(TEAM_WAVES, kt, lds_t)
                HIP_OK(h, hipGetLastError());
                if (!h->capturing) HIP_OK(h, hipEventRecord(h->ev[3], s));
                h->timing_valid = !h->capturing;
                team_done(kt, lds_t, TEAM_WAVES);
                return 0;
            }
        }
        // (B) up to four problems per compute unit (or a reservation): the compaction runs as its own launch, the tables are sized
        //     from the batch's maxima (read back: microseconds against a solve of tens of milliseconds) or from the reservation, and
        //     a problem gets TWO wavefronts when four such workgroups fit a compute unit -- the whole batch is resident at once.
        if (int r = prepare(h, B, p, s, io, true, trk, refs)) return r;
        prepared = true;
        io.u0 = u0; io.y0 = y0; io.c0 = c0; io.u = u; io.cost = cost; io.status = status; io.inner_it = inner_it;
        io.outer_it = outer_it; io.fpr = fpr; io.f2norm = f2norm; io.y_out = y_out; io.ms = ms;
        io.evals = (int32_t*)h->evals.ptr;
        const int tw = h->yield_waves ? h->yield_waves : (B <= 2 * h->num_cus ? TEAM_WAVES : 2);   // (MPCGPU_OPT_TAIL_WAVES: a test knob here)
        KParams kt = h->kp;
        fill_team_layout(kt, tw, h->kp.mKs, h->kp.mKf, h->kp.mKd);
        kt.reserved = 1;                       // every problem is checked against the tables' size on the device
        const size_t lds_t = kt.l_total * sizeof(double);
        const int per_cu = 4 * MPC_TEAM_WPE / tw;   // 256 VGPRs per wavefront: eight wavefronts per compute unit
        if ((lds_t + PREP_STATIC_LDS) * per_cu <= 160 * 1024) {
            io.p = nullptr;                    // the records exist already
            if (!h->capturing) HIP_OK(h, hipEventRecord(h->ev[2], s));
            if (tw == 2) { LAUNCH_TEAM_N(2, kt, lds_t) } else { LAUNCH_TEAM_N(TEAM_WAVES, kt, lds_t) }
            HIP_OK(h, hipGetLastError());
            if (!h->capturing) HIP_OK(h, hipEventRecord(h->ev[3], s));
            h->timing_valid = !h->capturing;
            team_done(kt, lds_t, tw);
            return 0;
        }
    }
#undef LAUNCH_TEAM_N
#undef LAUNCH_TEAM
    if (!prepared)
        if (int r = prepare(h, B, p, s, io, true, trk, refs)) return r;
    io.u0 = u0; io.y0 = y0; io.c0 = c0; io.u = u; io.cost = cost; io.status = status; io.inner_it = inner_it;
    io.outer_it = outer_it; io.fpr = fpr; io.f2norm = f2norm; io.y_out = y_out; io.ms = ms;
    if (int r = ensure(h, h->evals, (size_t)B * 2 * sizeof(int32_t))) return r;
    io.evals = (int32_t*)h->evals.ptr;
    h->last_B = B;
    if (!h->capturing) HIP_OK(h, hipEventRecord(h->ev[2], s));
    // N_hor = 40 (round 4): when every active dynamic row of the batch is an axis-aligned disc on a straight line (the reference's
    // own prediction feeder, src/main.py:77-85), the centres travel as linear tables (prep_problem: lossless, 0.9 instead of 5 KB)
    // and the carve of the 128-register build -- stash included -- fits a compute unit 16 times: four wavefronts per SIMD instead
    // of three.  Taken when the batch has more problems than the three-wavefront build keeps resident.
    bool lin40 = false;
    h->last_linear = false;
    KParams kp_stored = h->kp;    // the layout with the stored centres: what the pick-up launch of a linear-table launch runs with
    if (MPC_LINEAR40 && compiled_horizon(h) == 40 && h->opt_linear && h->shape_const && h->axis_aligned && !h->last_pairing &&
        LBFGS_IN_WORKSPACE && MPC_TRY_FOUR_WAVES && B > 4 * MPC_MIN_WAVES * h->num_cus) {
        fill_lds_layout(h->kp, h->kp.mKs, h->kp.mKf, h->kp.mKd, true, false, true, 4);
        if (h->kp.l_total * (int)sizeof(double) <= (MPC_STW40_W4 == 6 ? 9 * 1280 : 10 * 1024)) lin40 = true;
        else fill_lds_layout(h->kp, h->kp.mKs, h->kp.mKf, h->kp.mKd, true, !LBFGS_IN_WORKSPACE);
        h->last_shape[3] = h->kp.l_total * (int)sizeof(double);
    }
    const size_t lds = h->kp.l_total * sizeof(double);
    // more than 64 KiB of dynamic LDS (long horizons with many time-varying obstacles) must be opted into per kernel
#define LAUNCH_PAIR_WA(NT, SC, MINW, AX)                                                                           \
    do {                                                                                                             \
        auto kern = solve_kernel_pair<NT, SC, LBFGS_IN_WORKSPACE, MINW, AX>;                                         \
        if (int r_ = opt_in_lds(h, (const void*)kern, lds)) return r_;                                               \
        hipLaunchKernelGGL(kern, dim3(B), dim3(WAVE), lds, s, h->kp, io, B);                                         \
    } while (0)
    // compile-time horizons for the configurations the reference uses (generic kernel otherwise) x
    // {shape-constant, general} dynamic-obstacle tables
    // Residency: 160 KiB of LDS and 4 x 512 VGPRs per CU.  When a wavefront's LDS carve fits 16 times (<= 10 KiB) AND
    // the batch has more problems than the 148-VGPR build can keep resident (12 per CU), the 128-VGPR build (4
    // wavefronts per SIMD, ~20 spilled registers) wins by 9-12 %; a batch that fits anyway, or a bigger carve, runs
    // the build without spills (it is 13 % faster per wavefront).
    // shape-constant tables whose rows are all axis-aligned (the reference's own prediction feeder): the kernel without the
    // rotation into the ellipse frame -- the same bits, fewer instructions
#define LAUNCH_PAIR_W(NT, SC, MINW)                                                                                \
    do {                                                                                                             \
        if ((SC) && h->axis_aligned) LAUNCH_PAIR_WA(NT, SC, MINW, (SC)); else LAUNCH_PAIR_WA(NT, SC, MINW, false);  \
    } while (0)
#define LAUNCH_PAIR(NT, SC) LAUNCH_PAIR_W(NT, SC, MPC_MIN_WAVES)
    const bool sc = h->shape_const;
    const bool four = MPC_TRY_FOUR_WAVES && lds <= 10 * 1024 && B > MPC_FOUR_WAVES_FROM * h->num_cus;
    h->last_min_waves = (four && compiled_horizon(h) == 20) || lin40 ? 4 : MPC_MIN_WAVES;
    // MPCGPU_OPT_ORDER (mpc_order.hpp): longest first by the evaluation counts the previous call of this batch size left in
    // `evals` -- only when the batch is larger than what is resident at once (else everything starts together anyway).  The three
    // small kernels are part of the timed solve.
    io.perm = nullptr;
    h->last_ordered = 0;
    int resident = 0;
    {   // wavefronts resident at once: registers (4 per SIMD for the 128-VGPR build, else MPC_MIN_WAVES) and the LDS carve
        const size_t granule = 1280, wg_lds = (lds * (h->last_pairing ? 2 : 1) + granule - 1) / granule * granule;   // LDS is handed out in 1280-byte granules
        const int by_regs = 4 * (h->last_pairing ? 2 : h->last_min_waves), by_lds = (int)(160 * 1024 / wg_lds);
        resident = (by_regs < by_lds ? by_regs : by_lds) * h->num_cus * (h->last_pairing ? 2 : 1);
    }
    if (h->order == 1 && h->evals_B == B && B > resident && same_stream_as_last) {
        if (int r = ensure(h, h->perm, (size_t)B * sizeof(int32_t))) return r;
        if (int r = ensure(h, h->bins, ORD_BINS * sizeof(int))) return r;
        // bin width: 1024 bins over the largest possible evaluation count (about 12 per PANOC step)
        int shift = 0;
        while (((long long)h->kp.max_inner * h->kp.max_outer * 12 + 64) >> shift > ORD_BINS) ++shift;
        HIP_OK(h, hipMemsetAsync(h->bins.ptr, 0, ORD_BINS * sizeof(int), s));
        hipLaunchKernelGGL(order_hist_kernel, dim3((B + 255) / 256), dim3(256), 0, s, (const int32_t*)h->evals.ptr, B, (int*)h->bins.ptr, shift);
        hipLaunchKernelGGL(order_scan_kernel, dim3(1), dim3(ORD_BINS), 0, s, (int*)h->bins.ptr);
        hipLaunchKernelGGL(order_scatter_kernel, dim3((B + 255) / 256), dim3(256), 0, s, (const int32_t*)h->evals.ptr, B, (int*)h->bins.ptr,
                           (int32_t*)h->perm.ptr, shift);
        io.perm = (const int32_t*)h->perm.ptr;
        h->last_ordered = 1;
    }
    h->evals_B = B;
    // Tail promotion (mpc_kernels.hpp YIELD): once all but K problems of this launch have finished, the wavefronts that are still
    // running leave at the start of their next inner problem and a continuation launch of the latency kernel (four wavefronts per
    // problem) finishes them -- bitwise the same results.
    int yield_K = 0, yield_tw = 0;
    KParams kt_y{};
    size_t lds_y = 0;
    // Also on launches that start their problems longest first: with PERFECT hints (a bench step that repeats its batch) the tail is
    // short already and the continuation costs 1 % (210 -> 213 ms at B = 8192), but with the REAL hints of a receding-horizon loop
    // the two add up (8192 robots, cold: 168 ms per tick plain, 159 promoted, 160 ordered, 152 both: profiles/r05_closed_loop_ab.txt).
    if (MPC_STEP_LOOP && h->yield_opt != 0 && !h->last_pairing && !lin40) {
        // How many problems of the latency kernel a compute unit holds: 8 wavefronts of that kernel by registers, and the LDS carve
        // (tables for this batch's maxima).
        auto team_shape = [&](int tw, KParams& kt, size_t& lds_t) -> int {
            kt = h->kp;
            fill_team_layout(kt, tw, h->kp.mKs, h->kp.mKf, h->kp.mKd);
            kt.reserved = 1;
            lds_t = kt.l_total * sizeof(double);
            const int by_lds = (int)((160 * 1024) / (lds_t + PREP_STATIC_LDS)), by_regs = 4 * MPC_TEAM_WPE / tw;
            return by_lds < by_regs ? by_lds : by_regs;
        };
        int tw = h->yield_waves ? h->yield_waves : TEAM_WAVES;
        int per_cu = team_shape(tw, kt_y, lds_y);
        // Automatic capacity: TWICE what is resident at once when a compute unit holds two teams (N_hor = 20: 1024 -- the second
        // half starts as the first finishes; 252.7 -> 248.4 ms at B = 8192, profiles/r05_tail_promotion_ab.txt), else what is
        // resident (N_hor = 40: 256).  An explicit capacity beyond four times the residency takes two wavefronts per problem.
        // With the continuation running while the launch drains (the default outside a capture) the teams take over as they come
        // free: twice the residency in either case (N_hor = 40: 512 -- 294.7 -> 280 ms at B = 4096, profiles/r05_tail_concurrent_ab.txt).
        const bool may_overlap = h->tail_concurrent && !h->capturing;
        int K = h->yield_opt > 0 ? h->yield_opt : (per_cu >= 2 || may_overlap ? 2 : 1) * per_cu * h->num_cus;
        if (!h->yield_waves && h->yield_opt > 4 * per_cu * h->num_cus) { tw = 2; per_cu = team_shape(tw, kt_y, lds_y); }
        // a small batch: no more than 9/16 of it (B = 1280: 93.7 ms with 768 against 99.6 with 1024; from 2048 problems on the
        // automatic capacity is below that anyway: profiles/r05_tail_concurrent_ab.txt, block 7)
        if (h->yield_opt < 0 && B > per_cu * h->num_cus && K > B * 9 / 16) K = B * 9 / 16;   // (a batch the teams hold at once: all of it)
        if (K > B) K = B;
        if (per_cu >= 1 && K > 0) {
            if (int r = ensure(h, h->ylist, (size_t)K * sizeof(int32_t))) return r;
            yield_K = K; yield_tw = tw;
            h->kp.yield_from = B - K > 0 ? B - K : 1;
            h->kp.yield_cap = K;
            h->kp.yield_mask = h->yield_poll - 1;
            io.ylist = (int32_t*)h->ylist.ptr;
            kt_y.yield_cap = K;
            kt_y.yield_mask = h->yield_poll - 1;
            kt_y.yield_total = B;
            // every entry starts out EMPTY (-1): the concurrent continuation waits for an entry to appear, the sweep skips what is not >= 0
            HIP_OK(h, hipMemsetAsync(h->ylist.ptr, 0xFF, (size_t)K * sizeof(int32_t), s));
        }
    }
    // Concurrent continuation (MPCGPU_OPT_TAIL_CONCURRENT): the latency kernel on the side stream, behind a gate that opens when the
    // throughput launch starts to promote; its workgroups take list entries as they appear.  Not while capturing (the launch behind
    // the throughput kernel is what a graph records), and only when every unfinished problem is resident by the time the gate opens
    // (K <= what the throughput kernel holds at once): workgroups of the latency kernel must never wait for list entries while they
    // keep problems of the launch from starting.  For the same reason not while ANOTHER handle's launch is in flight on this device
    // (two streams kept busy by two handles: that launch fills the drain of this one, and the teams would wait for registers it
    // holds -- 3.37e4 -> 2.91e4 solves/s measured).
    const bool concurrent = yield_K > 0 && h->tail_concurrent && !h->capturing && yield_K <= resident && !another_launch_in_flight(h);
    if (concurrent) {
        h->kp.yield_grad = h->tail_gradual;                  // gradual promotion needs the teams beside the launch
        h->kp.yield_total = B;
        HIP_OK(h, hipEventRecord(h->ev_fork, s));            // the records (compaction) and the empty list are in place
        HIP_OK(h, hipStreamWaitEvent(h->side, h->ev_fork, 0));
    }
#define LAUNCH_DUO(NT, SC)                                                                                          \
    do {                                                                                                             \
        auto kern = solve_kernel_duo<NT, SC, LBFGS_IN_WORKSPACE>;                                                    \
        if (int r_ = opt_in_lds(h, (const void*)kern, 2 * lds)) return r_;                                           \
        hipLaunchKernelGGL(kern, dim3((B + 1) / 2), dim3(WAVE), 2 * lds, s, h->kp, io, B);                           \
    } while (0)
    if (h->last_pairing) {
        h->last_min_waves = 2;
        if (sc) LAUNCH_DUO(20, true); else LAUNCH_DUO(20, false);
    } else
    switch (compiled_horizon(h)) {
        case 20:
            if (four) { if (sc) LAUNCH_PAIR_W(20, true, 4); else LAUNCH_PAIR_W(20, false, 4); }
            else if (sc) LAUNCH_PAIR(20, true); else LAUNCH_PAIR(20, false);
            break;
        case 40:
#if MPC_LINEAR40
            if (lin40) {
                // Problems with a row that does not fit the linear tables (H_NLIN: curved predictions, a coordinate next to zero) are
                // skipped by this launch and listed for a pick-up launch of the stored-centre kernel right behind it: its grid is
                // the whole batch, workgroups beyond the (device-side) length of the list leave at once.  Nothing is read back.
                if (int r_ = ensure(h, h->nlist, (size_t)B * sizeof(int32_t))) return r_;
                hipLaunchKernelGGL(nl_list_kernel, dim3((B + 255) / 256), dim3(256), 0, s, (const double*)h->ws.ptr, h->kp.ws_stride, B, io.counts,
                                   (int32_t*)h->nlist.ptr);
                auto kern = solve_kernel_pair<40, true, LBFGS_IN_WORKSPACE, 4, true, true>;
                if (int r_ = opt_in_lds(h, (const void*)kern, lds)) return r_;
                hipLaunchKernelGGL(kern, dim3(B), dim3(WAVE), lds, s, h->kp, io, B);
                BatchPtrs io2 = io;
                io2.perm = (const int32_t*)h->nlist.ptr;
                io2.nsel = io.counts + CNT_NL_COUNT;
                auto kern2 = solve_kernel_pair<40, true, LBFGS_IN_WORKSPACE, MPC_MIN_WAVES, true, false>;
                const size_t lds2 = kp_stored.l_total * sizeof(double);
                if (int r_ = opt_in_lds(h, (const void*)kern2, lds2)) return r_;
                hipLaunchKernelGGL(kern2, dim3(B), dim3(WAVE), lds2, s, kp_stored, io2, B);
                h->last_linear = true;
            } else
#endif
            if (sc) LAUNCH_PAIR(40, true); else LAUNCH_PAIR(40, false);
            break;
        default: if (sc) LAUNCH_PAIR(0, true); else LAUNCH_PAIR(0, false); break;
    }
#undef LAUNCH_DUO
#undef LAUNCH_PAIR
#undef LAUNCH_PAIR_WA
#undef LAUNCH_PAIR_W
    HIP_OK(h, hipGetLastError());
    h->tail_timed = false;
    if (yield_K > 0) {
        if (!h->capturing) { HIP_OK(h, hipEventRecord(h->ev[4], s)); h->tail_timed = true; }
        // the continuation: grid = capacity of the list, workgroups beyond its device-side length leave at once
#define LAUNCH_RESUME(NT, TW, GRID, STREAM)                                                                          \
    do {                                                                                                             \
        auto kern = solve_kernel_team<NT, TW>;                                                                       \
        if (int r_ = opt_in_lds(h, (const void*)kern, lds_y)) return r_;                                             \
        hipLaunchKernelGGL(kern, dim3(GRID), dim3(WAVE * TW), lds_y, STREAM, kt_y, io, B);                           \
    } while (0)
#define LAUNCH_RESUME_N(GRID, STREAM)                                                                                \
    switch (compiled_horizon(h)) {                                                                                   \
        case 20: if (yield_tw == 2) LAUNCH_RESUME(20, 2, GRID, STREAM); else LAUNCH_RESUME(20, TEAM_WAVES, GRID, STREAM); break; \
        case 40: if (yield_tw == 2) LAUNCH_RESUME(40, 2, GRID, STREAM); else LAUNCH_RESUME(40, TEAM_WAVES, GRID, STREAM); break; \
        default: if (yield_tw == 2) LAUNCH_RESUME(0, 2, GRID, STREAM); else LAUNCH_RESUME(0, TEAM_WAVES, GRID, STREAM); break;   \
    }
        io.p = nullptr; io.perm = nullptr;
        if (concurrent) {
            // side stream: gate (opens at FINISHED >= yield_from and STARTED = B; 60 s limit), then one workgroup per list entry
            hipLaunchKernelGGL(tail_gate_kernel, dim3(1), dim3(WAVE), 0, h->side, (int*)io.counts, h->kp.yield_from, B, 6000000000LL, h->kp.yield_grad);
            kt_y.yield_persist = 1;
            LAUNCH_RESUME_N(yield_K, h->side)
            HIP_OK(h, hipGetLastError());
            HIP_OK(h, hipEventRecord(h->ev_join, h->side));
            HIP_OK(h, hipStreamWaitEvent(s, h->ev_join, 0));
            kt_y.yield_persist = 0;   // the sweep below: entries the side launch did not get to (normally none)
            h->last_concurrent = true;
        }
        LAUNCH_RESUME_N(yield_K, s)
#undef LAUNCH_RESUME_N
#undef LAUNCH_RESUME
        HIP_OK(h, hipGetLastError());
        h->last_yield_cap = yield_K;
    }
    if (!h->capturing) HIP_OK(h, hipEventRecord(h->ev[3], s));
    h->timing_valid = !h->capturing;
    return 0;
}
}  // namespace

extern "C" {

int32_t mpcgpu_solve_batch_dev(void* handle, int32_t B, const double* p, const double* u0, const double* y0,
                               const double* c0, double* u, double* cost, int32_t* status, int32_t* inner_it,
                               int32_t* outer_it, double* fpr, double* f2norm, double* y_out, double* ms,
                               void* stream) {
    Handle* h = (Handle*)handle;
    if (!h) return -1;
    if (B < 0) return fail(h, -1, "B=%d is negative", B);
    if (B == 0) return 0;
    if (!p || !u || !cost || !status) return fail(h, -1, "p, u, cost and status must not be NULL");
    HIP_OK(h, hipSetDevice(h->device));
    return solve_common(h, B, p, nullptr, nullptr, u0, y0, c0, u, cost, status, inner_it, outer_it, fpr, f2norm, y_out, ms,
                        pick_stream(h, stream));
}

// ---- batched tracker harness on the device (mpc_tracker.hpp) -------------------------------------------------------------
namespace {
int check_tracker(Handle* h, const mpcgpu_tracker* t) {
    if (!t) return fail(h, -1, "tracker is NULL");
    if (t->B < 0 || t->ref_cap < 1 || t->action_steps < 1 || t->action_steps > h->kp.N)
        return fail(h, -1, "tracker: B=%d ref_cap=%d action_steps=%d", t->B, t->ref_cap, t->action_steps);
    if (!t->states || !t->goals || !t->last_actions || !t->ref || !t->ref_len || !t->idx_ref || !t->stc || !t->dyn || !t->pred_states || !t->active)
        return fail(h, -1, "tracker: only `other` may be NULL");
    return 0;
}
}  // namespace

int32_t mpcgpu_tracker_window_dev(void* handle, const mpcgpu_tracker* t, double* refs_out, void* stream) {
    Handle* h = (Handle*)handle;
    if (!h) return -1;
    if (int r = check_tracker(h, t)) return r;
    if (!refs_out) return fail(h, -1, "refs_out is NULL");
    if (t->B == 0) return 0;
    HIP_OK(h, hipSetDevice(h->device));
    HIP_OK(h, launch_tracker_window(tracker_view(h, t), h->kp.N, refs_out, pick_stream(h, stream)));
    return 0;
}

int32_t mpcgpu_tracker_step_dev(void* handle, const mpcgpu_tracker* t, const double* refs, const double* u0, double* u,
                                double* cost, int32_t* status, int32_t* inner_it, int32_t* outer_it, double* actions_out,
                                void* stream) {
    Handle* h = (Handle*)handle;
    if (!h) return -1;
    if (int r = check_tracker(h, t)) return r;
    if (!refs || !u || !cost || !status) return fail(h, -1, "refs, u, cost and status must not be NULL");
    if (t->B == 0) return 0;
    HIP_OK(h, hipSetDevice(h->device));
    hipStream_t s = pick_stream(h, stream);
    if (int r = solve_common(h, t->B, nullptr, t, refs, u0, nullptr, nullptr, u, cost, status, inner_it, outer_it, nullptr, nullptr,
                             nullptr, nullptr, s)) return r;
    HIP_OK(h, launch_tracker_apply(tracker_view(h, t), h->kp.N, h->kp.ts, (const double*)u, actions_out, s));
    return 0;
}

int32_t mpcgpu_rl_reference_dev(void* handle, int32_t B, const double* agent, int32_t agent_stride, const int64_t* action,
                                double ts, int32_t steps, double ref_speed, const double* limits, double* rl_ref, void* stream) {
    Handle* h = (Handle*)handle;
    if (!h) return -1;
    if (B < 0 || steps < 1 || agent_stride < 5 || !agent || !action || !limits || !rl_ref) return fail(h, -1, "rl_reference: bad arguments");
    if (B == 0) return 0;
    HIP_OK(h, hipSetDevice(h->device));
    RlLimits lim{limits[0], limits[1], limits[2], limits[3], limits[4], limits[5], limits[6], limits[7]};
    HIP_OK(h, launch_rl_reference(B, agent, agent_stride, action, ts, steps, ref_speed, lim, rl_ref, pick_stream(h, stream)));
    return 0;
}

int32_t mpcgpu_hint_switch_dev(void* handle, int32_t B, int32_t N, int32_t O, int32_t V, const double* polygons,
                               const uint8_t* valid, const double* states, const double* original, const double* rl_ref,
                               int32_t rl_steps, const uint8_t* live, double switch_distance, double detach_distance,
                               double detach_steps, uint8_t* switch_on, int32_t* detach_cnt, double* chosen, void* stream) {
    Handle* h = (Handle*)handle;
    if (!h) return -1;
    if (B < 0 || N < 1 || O < 0 || V < 1 || !states || !original || !rl_ref || !switch_on || !detach_cnt || !chosen || (O > 0 && (!polygons || !valid)))
        return fail(h, -1, "hint_switch: bad arguments");
    if (B == 0) return 0;
    HIP_OK(h, hipSetDevice(h->device));
    HIP_OK(h, launch_hint_switch(B, N, O, V, polygons, valid, states, original, rl_ref, rl_steps, live, switch_distance, detach_distance,
                                 detach_steps, switch_on, detach_cnt, chosen, pick_stream(h, stream)));
    return 0;
}

// test hook: the tracker's assembly alone (fills the workspace records, nothing is solved)
int32_t mpcgpu_debug_tracker_assemble(void* handle, const mpcgpu_tracker* t, const double* refs) {
    Handle* h = (Handle*)handle;
    if (!h) return -1;
    if (int r = check_tracker(h, t)) return r;
    if (!refs) return fail(h, -1, "refs is NULL");
    HIP_OK(h, hipSetDevice(h->device));
    h->capturing = false;
    BatchPtrs io{};
    if (int r = prepare(h, t->B, nullptr, h->stream, io, false, t, refs)) return r;
    HIP_OK(h, hipStreamSynchronize(h->stream));
    return 0;
}

// test hook: the L-BFGS operator alone, in both forms (lbfgs_direction_kernel)
int32_t mpcgpu_debug_lbfgs_direction(void* handle, int32_t B, int32_t m, const double* U, const double* R, double* d_gram,
                                     double* d_twoloop, int32_t* pairs) {
    Handle* h = (Handle*)handle;
    if (!h || B <= 0 || m < 1 || !U || !R || !d_gram || !d_twoloop) return -1;
    const int NT = compiled_horizon(h);
    if (!NT || !gram_layout(h->kp)) return fail(h, -1, "the Gram form exists for the compiled horizons (N_hor = 20, 40; L-BFGS memory 10)");
    HIP_OK(h, hipSetDevice(h->device));
    h->capturing = false;
    const size_t n = 2 * (size_t)h->kp.N, seq = (size_t)B * (m + 1) * n * 8;
    DevBuf dU, dR;   // one-off buffers of a test hook
    if (int r = ensure(h, dU, seq)) return r;
    if (int r = ensure(h, dR, seq)) { (void)hipFree(dU.ptr); return r; }
    int rc = 0;
    do {
        if ((rc = ensure(h, h->ws, (size_t)B * h->kp.ws_stride * sizeof(double)))) break;
        if ((rc = ensure(h, h->u, (size_t)B * n * 8))) break;
        if ((rc = ensure(h, h->grad, (size_t)B * n * 8))) break;
        if ((rc = ensure(h, h->evals, (size_t)B * 2 * sizeof(int32_t)))) break;
        hipStream_t s = h->stream;
        if (hipMemcpyAsync(dU.ptr, U, seq, hipMemcpyHostToDevice, s) != hipSuccess || hipMemcpyAsync(dR.ptr, R, seq, hipMemcpyHostToDevice, s) != hipSuccess) { rc = fail(h, -10, "hipMemcpyAsync failed"); break; }
        const size_t lds = fixed_lds(NT, h->kp.mem, false).end * sizeof(double);
        if (NT == 20) hipLaunchKernelGGL(lbfgs_direction_kernel<20>, dim3(B), dim3(WAVE), lds, s, h->kp, (double*)h->ws.ptr, (const double*)dU.ptr, (const double*)dR.ptr, m,
                                         (double*)h->u.ptr, (double*)h->grad.ptr, (int32_t*)h->evals.ptr, B);
        else hipLaunchKernelGGL(lbfgs_direction_kernel<40>, dim3(B), dim3(WAVE), lds, s, h->kp, (double*)h->ws.ptr, (const double*)dU.ptr, (const double*)dR.ptr, m,
                                (double*)h->u.ptr, (double*)h->grad.ptr, (int32_t*)h->evals.ptr, B);
        if (hipGetLastError() != hipSuccess) { rc = fail(h, -10, "launch failed"); break; }
        if (hipMemcpyAsync(d_gram, h->u.ptr, (size_t)B * n * 8, hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipMemcpyAsync(d_twoloop, h->grad.ptr, (size_t)B * n * 8, hipMemcpyDeviceToHost, s) != hipSuccess ||
            (pairs && hipMemcpyAsync(pairs, h->evals.ptr, (size_t)B * 2 * sizeof(int32_t), hipMemcpyDeviceToHost, s) != hipSuccess) ||
            hipStreamSynchronize(s) != hipSuccess) { rc = fail(h, -10, "copy back failed"); break; }
        h->evals_B = 0; h->last_B = 0;   // the counters buffer was borrowed
    } while (0);
    (void)hipFree(dU.ptr); (void)hipFree(dR.ptr);
    return rc;
}

int32_t mpcgpu_workspace_stride(void* handle) { Handle* h = (Handle*)handle; return h ? h->kp.ws_stride : -1; }
int32_t mpcgpu_workspace_record(void* handle) { Handle* h = (Handle*)handle; return h ? h->kp.ws_lbs : -1; }

int32_t mpcgpu_debug_read_workspace(void* handle, int32_t B, double* out) {
    Handle* h = (Handle*)handle;
    if (!h || !out || B < 0) return -1;
    if ((size_t)B * h->kp.ws_stride * sizeof(double) > h->ws.cap) return fail(h, -4, "the workspace holds fewer than %d records", B);
    HIP_OK(h, hipSetDevice(h->device));
    HIP_OK(h, hipDeviceSynchronize());
    HIP_OK(h, hipMemcpy(out, h->ws.ptr, (size_t)B * h->kp.ws_stride * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

int32_t mpcgpu_debug_prep(void* handle, int32_t B, const double* p) {
    Handle* h = (Handle*)handle;
    if (!h || !p || B <= 0) return -1;
    HIP_OK(h, hipSetDevice(h->device));
    h->capturing = false;
    if (int r = ensure(h, h->p, (size_t)B * h->kp.np * 8)) return r;
    HIP_OK(h, hipMemcpyAsync(h->p.ptr, p, (size_t)B * h->kp.np * 8, hipMemcpyHostToDevice, h->stream));
    BatchPtrs io{};
    if (int r = prepare(h, B, (const double*)h->p.ptr, h->stream, io, false)) return r;
    HIP_OK(h, hipStreamSynchronize(h->stream));
    return 0;
}

int32_t mpcgpu_solve_batch(void* handle, int32_t B, const double* p, const double* u0, const double* y0,
                           const double* c0, double* u, double* cost, int32_t* status, int32_t* inner_it,
                           int32_t* outer_it, double* fpr, double* f2norm, double* y_out, double* ms) {
    Handle* h = (Handle*)handle;
    if (!h) return -1;
    if (B < 0) return fail(h, -1, "B=%d is negative", B);
    if (B == 0) return 0;
    if (!p || !u || !cost || !status) return fail(h, -1, "p, u, cost and status must not be NULL");
    HIP_OK(h, hipSetDevice(h->device));
    const size_t n = 2 * (size_t)h->kp.N, np = (size_t)h->kp.np, Bz = (size_t)B;
    hipStream_t s = h->stream;
    if (int r = ensure(h, h->p, Bz * np * 8)) return r;
    if (int r = ensure(h, h->u, Bz * n * 8)) return r;
    if (int r = ensure(h, h->cost, Bz * 8)) return r;
    if (int r = ensure(h, h->status, Bz * 4)) return r;
    if (int r = ensure(h, h->inner, Bz * 4)) return r;
    if (int r = ensure(h, h->outer, Bz * 4)) return r;
    if (int r = ensure(h, h->fpr, Bz * 8)) return r;
    if (int r = ensure(h, h->f2, Bz * 8)) return r;
    if (int r = ensure(h, h->y, Bz * n * 8)) return r;
    if (int r = ensure(h, h->ms, Bz * 8)) return r;
    HIP_OK(h, hipMemcpyAsync(h->p.ptr, p, Bz * np * 8, hipMemcpyHostToDevice, s));
    if (u0) { if (int r = ensure(h, h->u0, Bz * n * 8)) return r; HIP_OK(h, hipMemcpyAsync(h->u0.ptr, u0, Bz * n * 8, hipMemcpyHostToDevice, s)); }
    if (y0) { if (int r = ensure(h, h->y0, Bz * n * 8)) return r; HIP_OK(h, hipMemcpyAsync(h->y0.ptr, y0, Bz * n * 8, hipMemcpyHostToDevice, s)); }
    if (c0) { if (int r = ensure(h, h->c0, Bz * 8)) return r; HIP_OK(h, hipMemcpyAsync(h->c0.ptr, c0, Bz * 8, hipMemcpyHostToDevice, s)); }
    int r = mpcgpu_solve_batch_dev(h, B, (const double*)h->p.ptr, u0 ? (const double*)h->u0.ptr : nullptr,
                                   y0 ? (const double*)h->y0.ptr : nullptr, c0 ? (const double*)h->c0.ptr : nullptr,
                                   (double*)h->u.ptr, (double*)h->cost.ptr, (int32_t*)h->status.ptr,
                                   (int32_t*)h->inner.ptr, (int32_t*)h->outer.ptr, (double*)h->fpr.ptr,
                                   (double*)h->f2.ptr, (double*)h->y.ptr, (double*)h->ms.ptr, MPCGPU_STREAM_OWN);
    if (r) return r;
    HIP_OK(h, hipMemcpyAsync(u, h->u.ptr, Bz * n * 8, hipMemcpyDeviceToHost, s));
    HIP_OK(h, hipMemcpyAsync(cost, h->cost.ptr, Bz * 8, hipMemcpyDeviceToHost, s));
    HIP_OK(h, hipMemcpyAsync(status, h->status.ptr, Bz * 4, hipMemcpyDeviceToHost, s));
    if (inner_it) HIP_OK(h, hipMemcpyAsync(inner_it, h->inner.ptr, Bz * 4, hipMemcpyDeviceToHost, s));
    if (outer_it) HIP_OK(h, hipMemcpyAsync(outer_it, h->outer.ptr, Bz * 4, hipMemcpyDeviceToHost, s));
    if (fpr) HIP_OK(h, hipMemcpyAsync(fpr, h->fpr.ptr, Bz * 8, hipMemcpyDeviceToHost, s));
    if (f2norm) HIP_OK(h, hipMemcpyAsync(f2norm, h->f2.ptr, Bz * 8, hipMemcpyDeviceToHost, s));
    if (y_out) HIP_OK(h, hipMemcpyAsync(y_out, h->y.ptr, Bz * n * 8, hipMemcpyDeviceToHost, s));
    if (ms) HIP_OK(h, hipMemcpyAsync(ms, h->ms.ptr, Bz * 8, hipMemcpyDeviceToHost, s));
    HIP_OK(h, hipStreamSynchronize(s));
    return 0;
}

int32_t mpcgpu_cost_grad_batch(void* handle, int32_t B, const double* u, const double* xi, const double* p,
                               double* psi, double* f, double* grad, double* F1, double* F2) {
    Handle* h = (Handle*)handle;
    if (!h) return -1;
    if (B < 0) return fail(h, -1, "B=%d is negative", B);
    if (B == 0) return 0;
    if (!u || !xi || !p) return fail(h, -1, "u, xi and p must not be NULL");
    HIP_OK(h, hipSetDevice(h->device));
    const size_t n = 2 * (size_t)h->kp.N, np = (size_t)h->kp.np, Bz = (size_t)B, nd = (size_t)h->kp.Ndynobs;
    hipStream_t s = h->stream;
    if (int r = ensure(h, h->p, Bz * np * 8)) return r;
    if (int r = ensure(h, h->u, Bz * n * 8)) return r;
    if (int r = ensure(h, h->xi, Bz * (n + 1) * 8)) return r;
    if (int r = ensure(h, h->psi, Bz * 8)) return r;
    if (int r = ensure(h, h->f, Bz * 8)) return r;
    if (int r = ensure(h, h->grad, Bz * n * 8)) return r;
    if (int r = ensure(h, h->F1, Bz * n * 8)) return r;
    if (int r = ensure(h, h->F2, Bz * nd * 8)) return r;
    HIP_OK(h, hipMemcpyAsync(h->p.ptr, p, Bz * np * 8, hipMemcpyHostToDevice, s));
    HIP_OK(h, hipMemcpyAsync(h->u.ptr, u, Bz * n * 8, hipMemcpyHostToDevice, s));
    HIP_OK(h, hipMemcpyAsync(h->xi.ptr, xi, Bz * (n + 1) * 8, hipMemcpyHostToDevice, s));
    BatchPtrs io{};
    h->capturing = false;
    if (int r = prepare(h, B, (const double*)h->p.ptr, s, io, false)) return r;
    // test hook of the linear centre tables: the same evaluation through the LIN instantiation (N_hor = 40)
    const bool lin_cg = MPC_LINEAR40 && compiled_horizon(h) == 40 && h->opt_linear && h->shape_const && h->axis_aligned && h->linear && !h->last_pairing;
    if (lin_cg) fill_lds_layout(h->kp, h->kp.mKs, h->kp.mKf, h->kp.mKd, true, !LBFGS_IN_WORKSPACE, true, 3);
    h->last_linear = lin_cg;
    const size_t lds_cg = h->kp.l_total * sizeof(double);
#define LAUNCH_CGA(NT, SC, PP, GRID, LDSB, AX)                                                                      \
    do {                                                                                                             \
        auto kern = cost_grad_kernel<NT, SC, PP, AX>;                                                                \
        if (int r_ = opt_in_lds(h, (const void*)kern, (LDSB))) return r_;                                            \
        hipLaunchKernelGGL(kern, dim3(GRID), dim3(WAVE), (LDSB), s, h->kp, io, (const double*)h->u.ptr,              \
                           (const double*)h->xi.ptr, (double*)h->psi.ptr, (double*)h->f.ptr, (double*)h->grad.ptr,   \
                           (double*)h->F1.ptr, (double*)h->F2.ptr, B);                                               \
    } while (0)
#define LAUNCH_CG(NT, SC, PP, GRID, LDSB)                                                                           \
    do {                                                                                                             \
        if ((SC) && h->axis_aligned) LAUNCH_CGA(NT, SC, PP, GRID, LDSB, (SC)); else LAUNCH_CGA(NT, SC, PP, GRID, LDSB, false); \
    } while (0)
#define LAUNCH_CG1(NT, SC) LAUNCH_CG(NT, SC, Solo<NT>, B, lds_cg)
    if (h->last_pairing) {
        if (h->shape_const) LAUNCH_CG(20, true, Duo<20>, (B + 1) / 2, 2 * lds_cg);
        else LAUNCH_CG(20, false, Duo<20>, (B + 1) / 2, 2 * lds_cg);
    } else if (h->shape_const) {
        switch (compiled_horizon(h)) {
            case 20: LAUNCH_CG1(20, true); break;
            case 40:
#if MPC_LINEAR40
                if (lin_cg) {
                    auto kern = cost_grad_kernel<40, true, Solo<40>, true, true>;
                    if (int r_ = opt_in_lds(h, (const void*)kern, lds_cg)) return r_;
                    hipLaunchKernelGGL(kern, dim3(B), dim3(WAVE), lds_cg, s, h->kp, io, (const double*)h->u.ptr, (const double*)h->xi.ptr,
                                       (double*)h->psi.ptr, (double*)h->f.ptr, (double*)h->grad.ptr, (double*)h->F1.ptr, (double*)h->F2.ptr, B);
                } else
#endif
                LAUNCH_CG1(40, true);
                break;
            default: LAUNCH_CG1(0, true); break;
        }
    } else {
        switch (compiled_horizon(h)) {
            case 20: LAUNCH_CG1(20, false); break;
            case 40: LAUNCH_CG1(40, false); break;
            default: LAUNCH_CG1(0, false); break;
        }
    }
#undef LAUNCH_CG1
#undef LAUNCH_CG
#undef LAUNCH_CGA
#undef LAUNCH_CG
    HIP_OK(h, hipGetLastError());
    if (psi) HIP_OK(h, hipMemcpyAsync(psi, h->psi.ptr, Bz * 8, hipMemcpyDeviceToHost, s));
    if (f) HIP_OK(h, hipMemcpyAsync(f, h->f.ptr, Bz * 8, hipMemcpyDeviceToHost, s));
    if (grad) HIP_OK(h, hipMemcpyAsync(grad, h->grad.ptr, Bz * n * 8, hipMemcpyDeviceToHost, s));
    if (F1) HIP_OK(h, hipMemcpyAsync(F1, h->F1.ptr, Bz * n * 8, hipMemcpyDeviceToHost, s));
    if (F2) HIP_OK(h, hipMemcpyAsync(F2, h->F2.ptr, Bz * nd * 8, hipMemcpyDeviceToHost, s));
    HIP_OK(h, hipStreamSynchronize(s));
    return 0;
}

int32_t mpcgpu_last_timing(void* handle, double* prep_ms, double* solve_ms) {
    Handle* h = (Handle*)handle;
    if (!h) return -1;
    if (!h->timing_valid) return fail(h, -4, "no solve call has been timed yet");
    HIP_OK(h, hipSetDevice(h->device));
    HIP_OK(h, hipEventSynchronize(h->ev[3]));
    float a = 0.f, b = 0.f;
    HIP_OK(h, hipEventElapsedTime(&a, h->ev[0], h->ev[1]));
    HIP_OK(h, hipEventElapsedTime(&b, h->ev[2], h->ev[3]));
    if (prep_ms) *prep_ms = a;
    if (solve_ms) *solve_ms = b;
    return 0;
}

}  // extern "C"
namespace {
// A solve that was CAPTURED into a hipGraph leaves no completion event: its replays are ordered by the stream the graph is launched
// on, which only the caller knows.  Named stream: that stream alone is waited for (never the whole device: that would stall every
// other stream and fail outright while any of them is being captured).  MPCGPU_STREAM_OWN when the capture happened on ANOTHER stream
// names the wrong one -- waiting for it would hand back stale counters silently: the whole device is drained instead, which fails
// with a clear message while a capture is running anywhere on it.
int wait_for_captured_solve(Handle* h, void* stream_arg, hipStream_t s, const char* who) {
    if (stream_arg == MPCGPU_STREAM_OWN && h->last_stream != h->stream) {
        hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return fail(h, -6, "%s: the last solve was captured into a hipGraph on a stream of the caller's; pass the stream the graph is "
                               "launched on (draining the device instead failed: %s)", who, hipGetErrorString(e));
        }
        return 0;
    }
    if (stream_is_capturing(s)) return fail(h, -6, "%s inside a stream capture: the counters are read on the host", who);
    HIP_OK(h, hipStreamSynchronize(s));
    return 0;
}
}  // namespace
extern "C" {

int32_t mpcgpu_last_tail_timing(void* handle, double* main_ms, double* tail_ms) {
    Handle* h = (Handle*)handle;
    if (!h) return -1;
    if (!h->timing_valid) return fail(h, -4, "no solve call has been timed yet");
    HIP_OK(h, hipSetDevice(h->device));
    HIP_OK(h, hipEventSynchronize(h->ev[3]));
    float a = 0.f, b = 0.f;
    if (h->tail_timed) {
        HIP_OK(h, hipEventElapsedTime(&a, h->ev[2], h->ev[4]));
        HIP_OK(h, hipEventElapsedTime(&b, h->ev[4], h->ev[3]));
    } else {
        HIP_OK(h, hipEventElapsedTime(&a, h->ev[2], h->ev[3]));
    }
    if (main_ms) *main_ms = a;
    if (tail_ms) *tail_ms = b;
    return 0;
}

int32_t mpcgpu_last_eval_counts(void* handle, int32_t B, int32_t* n_psi, int32_t* n_grad, void* stream) {
    Handle* h = (Handle*)handle;
    if (!h) return -1;
    if (B != h->last_B || !h->evals.ptr) return fail(h, -4, "no solve of %d problems precedes this call (last: %d)", B, h->last_B);
    HIP_OK(h, hipSetDevice(h->device));
    // The counters are written by the solve kernel: wait for the EVENT recorded behind it (the launch stream is the caller's and may
    // have been destroyed since; a stream handle is never touched here unless the caller passes it now).  A solve that was captured
    // into a hipGraph has no such event: its replays are ordered by the stream the caller NAMES (the one the graph is launched on),
    // and that stream alone is waited for -- no device-wide wait, which would stall every other stream of the device and fail
    // outright while any of them is being captured.
    hipStream_t s = pick_stream(h, stream);
    if (!h->last_captured && h->timing_valid) {
        HIP_OK(h, hipEventSynchronize(h->ev[3]));
        if (s != h->last_stream) HIP_OK(h, hipStreamSynchronize(s));
    } else {
        if (int r = wait_for_captured_solve(h, stream, s, "mpcgpu_last_eval_counts")) return r;
    }
    int32_t* tmp = new (std::nothrow) int32_t[(size_t)B * 2];
    if (!tmp) return fail(h, -3, "out of host memory");
    hipError_t e = hipMemcpy(tmp, h->evals.ptr, (size_t)B * 2 * sizeof(int32_t), hipMemcpyDeviceToHost);
    if (e == hipSuccess)
        for (int i = 0; i < B; ++i) { if (n_psi) n_psi[i] = tmp[2 * i]; if (n_grad) n_grad[i] = tmp[2 * i + 1]; }
    delete[] tmp;
    if (e != hipSuccess) return fail(h, -10, "hipMemcpy failed: %s", hipGetErrorString(e));
    return 0;
}

int32_t mpcgpu_last_shape(void* handle, int32_t* max_static, int32_t* max_fleet, int32_t* max_dyn,
                          int32_t* lds_bytes) {
    Handle* h = (Handle*)handle;
    if (!h) return -1;
    if (max_static) *max_static = h->last_shape[0];
    if (max_fleet) *max_fleet = h->last_shape[1];
    if (max_dyn) *max_dyn = h->last_shape[2];
    if (lds_bytes) *lds_bytes = h->last_shape[3];
    return 0;
}

int32_t mpcgpu_reserve_shape(void* handle, int32_t max_static, int32_t max_fleet, int32_t max_dyn, int32_t var_shape) {
    Handle* h = (Handle*)handle;
    if (!h) return -1;
    if (max_static < 0 && max_fleet < 0 && max_dyn < 0) { h->reserved = false; return 0; }
    const mpcgpu_config& c = h->cfg;
    if (max_static < 0 || max_static > c.Nstcobs || max_fleet < 0 || max_fleet > c.Nother || max_dyn < 0 || max_dyn > c.Ndynobs)
        return fail(h, -1, "reserved shape (%d, %d, %d) outside the configured maxima (%d, %d, %d)", max_static, max_fleet,
                    max_dyn, c.Nstcobs, c.Nother, c.Ndynobs);
    KParams probe = h->kp;
    if (var_shape < 0 || var_shape > 2) return fail(h, -1, "var_shape must be 0 (shape-constant rows), 1 (rows may change shape) or 2 (shape-constant, axis-aligned), got %d", var_shape);
    fill_lds_layout(probe, max_static, max_fleet, max_dyn, var_shape != 1, !LBFGS_IN_WORKSPACE);
    if (probe.l_total * (int)sizeof(double) > 160 * 1024) return fail(h, -5, "reserved LDS carve of %d bytes exceeds 160 KiB", probe.l_total * 8);
    h->res_shape[0] = max_static; h->res_shape[1] = max_fleet; h->res_shape[2] = max_dyn; h->res_shape[3] = var_shape;
    h->reserved = true;
    return 0;
}

int32_t mpcgpu_reserve_batch(void* handle, int32_t B) {
    Handle* h = (Handle*)handle;
    if (!h) return -1;
    if (B < 0) return fail(h, -1, "B=%d is negative", B);
    HIP_OK(h, hipSetDevice(h->device));
    h->capturing = false;
    if (int r = ensure(h, h->ws, (size_t)B * h->kp.ws_stride * sizeof(double))) return r;
    if (int r = ensure(h, h->counts, CNT_WORDS * sizeof(int))) return r;
    if (int r = ensure(h, h->evals, (size_t)B * 2 * sizeof(int32_t))) return r;
    if (int r = ensure(h, h->perm, (size_t)B * sizeof(int32_t))) return r;
    if (int r = ensure(h, h->bins, ORD_BINS * sizeof(int))) return r;
    if (int r = ensure(h, h->nlist, (size_t)B * sizeof(int32_t))) return r;
    {   // tail promotion: the list of promoted problems, and the continuation kernel's LDS size when the carve is known (reservation)
        // (solve_common's K never exceeds its batch, whatever MPCGPU_OPT_TAIL_PROMOTION / MPCGPU_OPT_TAIL_WAVES / the team width of
        // the build make of it: one entry per problem of the promised batch covers every rule -- 4 bytes each)
        const int K = h->yield_opt > B ? h->yield_opt : B;
        if (int r = ensure(h, h->ylist, (size_t)(K > 0 ? K : 1) * sizeof(int32_t))) return r;
        if (h->reserved && h->yield_opt != 0) {
            for (int tw : {TEAM_WAVES, 2}) {
                KParams kt = h->kp;
                fill_team_layout(kt, tw, h->res_shape[0], h->res_shape[1], h->res_shape[2]);
                const size_t lds_t = kt.l_total * sizeof(double);
                if (lds_t + PREP_STATIC_LDS > 160 * 1024) continue;
                const int nt = compiled_horizon(h);
                const void* kern = tw == TEAM_WAVES ? (nt == 20 ? (const void*)solve_kernel_team<20, TEAM_WAVES> : nt == 40 ? (const void*)solve_kernel_team<40, TEAM_WAVES>
                                                                                                                     : (const void*)solve_kernel_team<0, TEAM_WAVES>)
                                                    : (nt == 20 ? (const void*)solve_kernel_team<20, 2> : nt == 40 ? (const void*)solve_kernel_team<40, 2>
                                                                                                                 : (const void*)solve_kernel_team<0, 2>);
                if (int r = opt_in_lds(h, kern, lds_t)) return r;
            }
        }
    }
    // A batch of the latency range is captured in its one-launch form (tables for the configured maxima: more than 64 KiB of
    // dynamic LDS for the yaml's slot counts), whatever form an eager call of the same size takes: opt that kernel in now.
    const int team_cap = h->team_max_batch >= 0 ? h->team_max_batch : 2 * h->num_cus;
    if (B <= team_cap && !use_duo(h) && !h->reserved) {
        KParams kt = h->kp;
        fill_team_layout(kt, TEAM_WAVES, h->cfg.Nstcobs, h->cfg.Nother, h->cfg.Ndynobs);
        const size_t lds_t = kt.l_total * sizeof(double);
        if (lds_t + PREP_STATIC_LDS <= 160 * 1024) {
            const void* kern = compiled_horizon(h) == 20 ? (const void*)solve_kernel_team<20, TEAM_WAVES>
                             : compiled_horizon(h) == 40 ? (const void*)solve_kernel_team<40, TEAM_WAVES>
                                                         : (const void*)solve_kernel_team<0, TEAM_WAVES>;
            if (int r = opt_in_lds(h, kern, lds_t)) return r;
        }
    }
    return 0;
}

int32_t mpcgpu_set_option(void* handle, int32_t option, double value) {
    Handle* h = (Handle*)handle;
    if (!h) return -1;
    switch (option) {
        case MPCGPU_OPT_LINESEARCH_FALLBACK:
            if (value != 0.0 && value != 1.0) return fail(h, -1, "linesearch fallback must be 0 (last trial) or 1 (tau = 0), got %g", value);
            h->kp.ls_fallback = (int)value;
            return 0;
        case MPCGPU_OPT_PENALTY_STALL:
            if (value != 0.0 && value != 1.0) return fail(h, -1, "penalty stall rule must be 0 (either infeasibility shrank) or 1 (both), got %g", value);
            h->kp.stall_rule = (int)value;
            return 0;
        case MPCGPU_OPT_TEAM_BATCH:
            if (value < -1.0 || value != (double)(int)value) return fail(h, -1, "team batch must be -1 (automatic) or a batch size >= 0, got %g", value);
            h->team_max_batch = (int)value;
            return 0;
        case MPCGPU_OPT_PAIRING:
            if (value != -1.0 && value != 0.0 && value != 1.0) return fail(h, -1, "pairing must be -1 (automatic), 0 or 1, got %g", value);
            if (value == 1.0 && !duo_available(h)) return fail(h, -1, "two problems per wavefront are compiled for N_hor = 20 only (N_hor = %d)", h->kp.N);
            h->pairing = (int)value;
            return 0;
        case MPCGPU_OPT_LINEAR_TABLES:
            if (value != 0.0 && value != 1.0) return fail(h, -1, "linear tables must be 0 (off) or 1 (on where compiled in), got %g", value);
            h->opt_linear = (int)value;   // has an effect only in builds with -DMPC_LINEAR40=1 (the variant libmpcgpu_linear40.so)
            return 0;
        case MPCGPU_OPT_ORDER:
            if (value != 0.0 && value != 1.0) return fail(h, -1, "order must be 0 (as given) or 1 (longest first by the previous call), got %g", value);
            h->order = (int)value;
            return 0;
        case MPCGPU_OPT_TAIL_PROMOTION:
            if (value < -1.0 || value != (double)(int)value) return fail(h, -1, "tail promotion must be -1 (automatic), 0 (off) or a number of problems, got %g", value);
            h->yield_opt = (int)value;
            return 0;
        case MPCGPU_OPT_TAIL_WAVES:
            if (value != 0.0 && value != 2.0 && value != 4.0) return fail(h, -1, "tail waves must be 0 (automatic), 2 or 4, got %g", value);
            h->yield_waves = (int)value;
            return 0;
        case MPCGPU_OPT_TAIL_CONCURRENT:
            if (value != 0.0 && value != 1.0) return fail(h, -1, "tail concurrent must be 0 or 1, got %g", value);
            h->tail_concurrent = (int)value;
            return 0;
        case MPCGPU_OPT_TAIL_GRADUAL:
            if (value < 0.0 || value > 1024.0 || value != (double)(int)value) return fail(h, -1, "tail gradual must be 0 (off) or finished problems per promoted one (1..1024), got %g", value);
            h->tail_gradual = (int)value;
            return 0;
        case MPCGPU_OPT_TAIL_POLL: {
            const int v = (int)value;
            if (value != (double)v || v < 1 || v > 4096 || (v & (v - 1)) != 0) return fail(h, -1, "tail poll interval must be a power of two in 1..4096, got %g", value);
            h->yield_poll = v;
            return 0;
        }
        default:
            return fail(h, -1, "unknown option %d", option);
    }
}

#ifdef MPC_TRACE
// trace builds only (tests): record the first `cap` PANOC steps of every problem of the following solves ...
int32_t mpcgpu_debug_set_trace(void* handle, int32_t cap) {
    Handle* h = (Handle*)handle;
    if (!h || cap < 0) return -1;
    h->trace_cap = cap;
    return 0;
}
// ... and copy the records of the last solve of B problems to the host: out[B][cap][12], NaN = not written
int32_t mpcgpu_debug_read_trace(void* handle, int32_t B, double* out) {
    Handle* h = (Handle*)handle;
    if (!h || !out) return -1;
    if (B != h->last_B || h->trace_cap <= 0 || !h->trace.ptr) return fail(h, -4, "no traced solve of %d problems precedes this call", B);
    HIP_OK(h, hipSetDevice(h->device));
    HIP_OK(h, hipDeviceSynchronize());
    HIP_OK(h, hipMemcpy(out, h->trace.ptr, (size_t)B * h->trace_cap * TRACE_W * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}
#endif

int32_t mpcgpu_last_waves_per_simd(void* handle) {
    Handle* h = (Handle*)handle;
    return h ? h->last_min_waves : -1;
}

int32_t mpcgpu_last_latency_kernel(void* handle) {
    Handle* h = (Handle*)handle;
    return h ? h->last_team : -1;   // 0: throughput kernel; 4 / 2: wavefronts per problem of the latency kernel
}

int32_t mpcgpu_last_table_kind(void* handle) {
    Handle* h = (Handle*)handle;
    if (!h) return -1;
    if (h->last_team) return 1;  // the latency kernel always carries the general tables
    if (h->last_linear) return 3;
    return h->shape_const ? (h->axis_aligned ? 2 : 0) : 1;
}

int32_t mpcgpu_last_problems_per_wavefront(void* handle) {
    Handle* h = (Handle*)handle;
    return h ? 1 + h->last_pairing : -1;
}
int32_t mpcgpu_last_ordered(void* handle) { Handle* h = (Handle*)handle; return h ? h->last_ordered : -1; }

int32_t mpcgpu_last_tail_promotion(void* handle, int32_t* promoted, void* stream) {
    Handle* h = (Handle*)handle;
    if (!h) return -1;
    if (promoted) {
        *promoted = 0;
        if (h->last_yield_cap > 0 && h->counts.ptr) {
            HIP_OK(h, hipSetDevice(h->device));
            hipStream_t s = pick_stream(h, stream);
            if (h->last_captured) { if (int r = wait_for_captured_solve(h, stream, s, "mpcgpu_last_tail_promotion")) return r; }
            else HIP_OK(h, hipStreamSynchronize(s));
            int n = 0;
            HIP_OK(h, hipMemcpy(&n, (const int*)h->counts.ptr + CNT_YIELDED, sizeof(int), hipMemcpyDeviceToHost));
            *promoted = n < h->last_yield_cap ? n : h->last_yield_cap;
        }
    }
    return h->last_yield_cap;
}

int32_t mpcgpu_last_tail_timeouts(void* handle, int32_t* timeouts, void* stream) {
    Handle* h = (Handle*)handle;
    if (!h || !timeouts) return -1;
    *timeouts = 0;
    if (h->last_yield_cap > 0 && h->counts.ptr) {
        HIP_OK(h, hipSetDevice(h->device));
        hipStream_t s = pick_stream(h, stream);
        if (h->last_captured) { if (int r = wait_for_captured_solve(h, stream, s, "mpcgpu_last_tail_timeouts")) return r; }
        else HIP_OK(h, hipStreamSynchronize(s));
        int n = 0;
        HIP_OK(h, hipMemcpy(&n, (const int*)h->counts.ptr + CNT_TIMEOUTS, sizeof(int), hipMemcpyDeviceToHost));
        *timeouts = n;
    }
    return h->last_concurrent ? 1 : 0;
}

#ifdef MPC_PROFILE
// profiling builds only: read and clear the phase-cycle table (24 counters)
int32_t mpcgpu_debug_read_prof(double* out24) {
    unsigned long long hbuf[NPROF];
    if (hipMemcpyFromSymbol(hbuf, HIP_SYMBOL(g_prof), sizeof(hbuf)) != hipSuccess) return -1;
    for (int i = 0; i < NPROF; ++i) out24[i] = (double)hbuf[i];
    unsigned long long z[NPROF] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof(z)) != hipSuccess) return -1;
    return 0;
}
#endif

}  // extern "C"
