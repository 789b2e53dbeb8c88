// trackgpu.hip -- the small per-robot kernels of the device tracker / hybrid loop (see mpc_tracker.hpp) and their launchers.
// Own translation unit, compiled without the ILP machine scheduler of the solver (it crashes LLVM on these rollout loops).
#include "mpc_tracker_types.hpp"

namespace mpcgpu {

constexpr int WAVE = 64;

// ---- get_local_ref_traj -------------------------------------------------------------------------------------------------
// one wavefront per robot; refs_out [B][N][3]
__global__ __launch_bounds__(WAVE) void tracker_window_kernel(TrackerView t, int N, double* __restrict__ refs_out) {
    const int b = blockIdx.x, lane = threadIdx.x;
    if (b >= t.B) return;
    const int a = t.action_steps, len = t.ref_len[b], idx = t.idx_ref[b];
    const double* ref = t.ref + (size_t)b * t.ref_cap * 3;
    const int lb = idx - a > 0 ? idx - a : 0;
    const int ub = idx + 5 * a < len ? idx + 5 * a : len;
    const double sx = t.states[3 * b], sy = t.states[3 * b + 1];
    // first minimum of the distances of the candidates lb .. ub-1 (ties: the first one, like list.index(min(d)))
    double best = __builtin_huge_val();
    int arg = 0;
    for (int j0 = 0; j0 < 6 * a; j0 += WAVE) {
        const int j = j0 + lane, cand = lb + j;
        double d = __builtin_huge_val();
        if (j < 6 * a && cand < ub) d = hypot(sx - ref[3 * cand], sy - ref[3 * cand + 1]);
        // wave minimum with the smallest lane on ties
        double m = d;
        int am = j;
        for (int off = 32; off >= 1; off >>= 1) {
            const double om = __shfl_xor(m, off);
            const int oa = __shfl_xor(am, off);
            if (om < m || (om == m && oa < am)) { m = om; am = oa; }
        }
        if (m < best) { best = m; arg = am; }
    }
    const int idx_next = lb + arg;
    if (lane == 0) t.idx_ref[b] = idx_next;
    for (int k = lane; k < N; k += WAVE) {
        const int r = idx_next + k < len - 1 ? idx_next + k : len - 1;
        double* o = refs_out + ((size_t)b * N + k) * 3;
        o[0] = ref[3 * r]; o[1] = ref[3 * r + 1]; o[2] = ref[3 * r + 2];
    }
}

// ---- post-solve part of run_step -------------------------------------------------------------------------------------
// literal four-stage RK4 of the unicycle (src/pkg_motion_model/motion_model.py:142-164), the order of the host's numpy form
__device__ __forceinline__ void unicycle_rk4(double& x, double& y, double& th, double v, double w, double ts) {
    double s, c;
    sincos(th, &s, &c);
    const double k1x = ts * (v * c), k1y = ts * (v * s), k1t = ts * w;
    sincos(th + 0.5 * k1t, &s, &c);
    const double k2x = ts * (v * c), k2y = ts * (v * s), k2t = ts * w;
    sincos(th + 0.5 * k2t, &s, &c);
    const double k3x = ts * (v * c), k3y = ts * (v * s), k3t = ts * w;
    sincos(th + k3t, &s, &c);
    const double k4x = ts * (v * c), k4y = ts * (v * s), k4t = ts * w;
    const double sixth = 1.0 / 6.0;
    const double dx = sixth * (k1x + 2.0 * k2x + 2.0 * k3x + k4x);   // separate statements: rounded like the host's numpy form
    const double dy = sixth * (k1y + 2.0 * k2y + 2.0 * k3y + k4y);
    const double dt = sixth * (k1t + 2.0 * k2t + 2.0 * k3t + k4t);
    x = x + dx; y = y + dy; th = th + dt;
}
// one thread per robot: u [B][N][2] the solution; actions_out [B][2] the applied first input (0 for robots that are done)
__global__ __launch_bounds__(128) void tracker_apply_kernel(TrackerView t, int N, double ts, const double* __restrict__ u, double* __restrict__ actions_out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= t.B) return;
    const double* ub = u + (size_t)b * 2 * N;
    const bool act = t.active[b] != 0;
    if (!act) {                                       // get_action returns None once the termination test has fired
        t.last_actions[2 * b] = 0.0; t.last_actions[2 * b + 1] = 0.0;
        if (actions_out) { actions_out[2 * b] = 0.0; actions_out[2 * b + 1] = 0.0; }
        return;
    }
    double x = t.states[3 * b], y = t.states[3 * b + 1], th = t.states[3 * b + 2];
    for (int s = 0; s < t.action_steps; ++s) unicycle_rk4(x, y, th, ub[2 * s], ub[2 * s + 1], ts);
    t.states[3 * b] = x; t.states[3 * b + 1] = y; t.states[3 * b + 2] = th;
    t.last_actions[2 * b] = ub[2 * (t.action_steps - 1)]; t.last_actions[2 * b + 1] = ub[2 * (t.action_steps - 1) + 1];
    if (actions_out) { actions_out[2 * b] = ub[0]; actions_out[2 * b + 1] = ub[1]; }
    // prediction: rolled from the TAKEN state with the whole input sequence again (the reference re-applies u[0]: kept)
    double* pr = t.pred_states + (size_t)b * N * 3;
    for (int k = 0; k < N; ++k) {
        unicycle_rk4(x, y, th, ub[2 * k], ub[2 * k + 1], ts);
        pr[3 * k] = x; pr[3 * k + 1] = y; pr[3 * k + 2] = th;
    }
}

// ---- the DQN's proposal (row f2) ------------------------------------------------------------------------------------------
// one thread per robot: agent [B][agent_stride] rows (x, y, theta, v, w, ...); action [B] in 0..8; rl_ref [B][steps][2]
__global__ __launch_bounds__(128) void rl_reference_kernel(int B, const double* __restrict__ agent, int agent_stride, const int64_t* __restrict__ action,
                                    double ts, int steps, double ref_speed, RlLimits lim, double* __restrict__ rl_ref) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double* s0 = agent + (size_t)b * agent_stride;
    double x = s0[0], y = s0[1], th = s0[2], v = s0[3], w = s0[4];
    const int a = (int)action[b];
    // step 0: the chosen acceleration pair (agent.py:124-145): index // 3 picks the linear, index % 3 the angular acceleration
    // Every product is a statement of its own: this file is built with -ffp-contract=on, which fuses an expression the way it is
    // written -- `th + ts * w` would become one FMA and the rollout would no longer be the host's numpy form (multiply, round, add).
    const double av0 = ts * (a / 3 == 0 ? lim.acc_max : 0.0), av1 = ts * (a / 3 == 2 ? lim.acc_min : 0.0);
    const double aw0 = ts * (a % 3 == 0 ? lim.angacc_max : 0.0), aw1 = ts * (a % 3 == 2 ? lim.angacc_min : 0.0);
    v = v + av0; v = v + av1;
    w = w + aw0; w = w + aw1;
    v = fmin(fmax(v, lim.speed_min), lim.speed_max);
    w = fmin(fmax(w, lim.angvel_min), lim.angvel_max);
    double sn, cs;
    const double dth0 = ts * w;
    th = th + dth0;
    sincos(th, &sn, &cs);
    const double tv = ts * v;
    const double dx0 = tv * cs, dy0 = tv * sn;
    x = x + dx0;
    y = y + dy0;
    double* o = rl_ref + (size_t)b * steps * 2;
    o[0] = x; o[1] = y;
    const double speed = ref_speed > 0.0 ? ref_speed : lim.speed_max;
    const double tsp = ts * speed;
#pragma unroll 1
    for (int j = 1; j < steps; ++j) {                 // agent.py:86-100: constant speed, turn rate decaying by 5 % per step
        w = w * 0.95;
        const double dth = ts * w;
        th = th + dth;
        sincos(th, &sn, &cs);
        const double dx = tsp * cs, dy = tsp * sn;
        x = x + dx;
        y = y + dy;
        o[2 * j] = x; o[2 * j + 1] = y;
    }
}

// even-odd rule, the form of the host's points_in_polygons
__device__ __forceinline__ bool point_in_ring(double px, double py, const double* ring, int V) {
    int crossings = 0;
    for (int i = 0; i < V; ++i) {
        const int j = i + 1 < V ? i + 1 : 0;
        const double ax = ring[2 * i], ay = ring[2 * i + 1], bx = ring[2 * j], by = ring[2 * j + 1];
        if ((ay > py) != (by > py)) {
            const double num = (bx - ax) * (py - ay);
            const double q = num / (by - ay);
            if (px < q + ax) ++crossings;
        }
    }
    return crossings & 1;
}
// shapely Polygon.distance(Point): 0 inside, else the distance to the outline
__device__ __forceinline__ double polygon_distance(double px, double py, const double* ring, int V) {
    if (point_in_ring(px, py, ring, V)) return 0.0;
    double best = __builtin_huge_val();
    for (int i = 0; i < V; ++i) {
        const int j = i + 1 < V ? i + 1 : 0;
        const double ax = ring[2 * i], ay = ring[2 * i + 1];
        const double dx = ring[2 * j] - ax, dy = ring[2 * j + 1] - ay;
        const double rx = px - ax, ry = py - ay;
        const double dxx = dx * dx, dyy = dy * dy;
        const double den = dxx + dyy;
        double tt = 0.0;
        if (den > 0.0) {
            const double n0 = rx * dx, n1 = ry * dy;
            tt = fmin(fmax((n0 + n1) / den, 0.0), 1.0);
        }
        const double tx = tt * dx, ty = tt * dy;
        const double gx = rx - tx, gy = ry - ty;
        const double g0 = gx * gx, g1 = gy * gy;
        best = fmin(best, sqrt(g0 + g1));
    }
    return best;
}
// one thread per robot.  polygons [B][O][V][2], valid [B][O] (uint8), positions = states [B][3] (x, y used), original [B][N][3],
// rl_ref [B][rl_steps][2]; switch_on [B] (uint8) and detach_cnt [B] (int32) are the switchers' state (in/out); live [B] (uint8,
// may be NULL): robots whose switcher is consulted this tick; chosen [B][N][3] = the reference to track.
__global__ __launch_bounds__(64) void hint_switch_kernel(int B, int N, int O, int V, const double* __restrict__ polygons, const uint8_t* __restrict__ valid,
                                   const double* __restrict__ states, const double* __restrict__ original,
                                   const double* __restrict__ rl_ref, int rl_steps, const uint8_t* __restrict__ live,
                                   double switch_distance, double detach_distance, double detach_steps,
                                   uint8_t* __restrict__ switch_on, int32_t* __restrict__ detach_cnt, double* __restrict__ chosen) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double* poly = polygons + (size_t)b * O * V * 2;
    const double* org = original + (size_t)b * N * 3;
    bool on = switch_on[b] != 0;
    int cnt = detach_cnt[b];
    const bool consulted = live == nullptr || live[b] != 0;
    if (consulted) {
        const double px = states[3 * b], py = states[3 * b + 1];
        bool counted = false, returned = false;
        // HintSwitcher.switch walks zip(original_traj, new_traj) (src/main_pre.py:37): min(N, rl_steps) rows -- with a 40-step
        // horizon and the 20-step proposal the last 20 rows of the original reference are never tested
        const int rows = N < rl_steps ? N : rl_steps;
        for (int r = 0; r < rows && !returned; ++r) {
            for (int o = 0; o < O && !returned; ++o) {
                if (!valid[(size_t)b * O + o]) continue;
                const double* ring = poly + (size_t)o * V * 2;
                const double dist = polygon_distance(px, py, ring, V);   // does not depend on the row (recomputed: O * N is small)
                if (point_in_ring(org[3 * r], org[3 * r + 1], ring, V)) {
                    if (dist < switch_distance && !on) { on = true; returned = true; }
                } else if (dist > detach_distance && on) {
                    if (cnt > detach_steps) { on = false; cnt = 0; }
                    else if (!counted) { ++cnt; counted = true; }
                }
            }
        }
        switch_on[b] = on ? 1 : 0;
        detach_cnt[b] = cnt;
    }
    const bool use = on && consulted;
    double* c = chosen + (size_t)b * N * 3;
    for (int k = 0; k < N; ++k) {
        const bool prop = use && k < rl_steps;
        c[3 * k] = prop ? rl_ref[((size_t)b * rl_steps + k) * 2] : org[3 * k];
        c[3 * k + 1] = prop ? rl_ref[((size_t)b * rl_steps + k) * 2 + 1] : org[3 * k + 1];
        c[3 * k + 2] = org[3 * k + 2];
    }
}


hipError_t launch_tracker_window(const TrackerView& t, int N, double* refs_out, hipStream_t s) {
    hipLaunchKernelGGL(tracker_window_kernel, dim3(t.B), dim3(WAVE), 0, s, t, N, refs_out);
    return hipGetLastError();
}
hipError_t launch_tracker_apply(const TrackerView& t, int N, double ts, const double* u, double* actions_out, hipStream_t s) {
    const int threads = 128;
    hipLaunchKernelGGL(tracker_apply_kernel, dim3((t.B + threads - 1) / threads), dim3(threads), 0, s, t, N, ts, u, actions_out);
    return hipGetLastError();
}
hipError_t launch_rl_reference(int B, const double* agent, int agent_stride, const int64_t* action, double ts, int steps,
                               double ref_speed, const RlLimits& lim, double* rl_ref, hipStream_t s) {
    const int threads = 128;
    hipLaunchKernelGGL(rl_reference_kernel, dim3((B + threads - 1) / threads), dim3(threads), 0, s, B, agent, agent_stride, action, ts,
                       steps, ref_speed, lim, rl_ref);
    return hipGetLastError();
}
hipError_t launch_hint_switch(int B, int N, int O, int V, const double* polygons, const uint8_t* valid, const double* states,
                              const double* original, const double* rl_ref, int rl_steps, const uint8_t* live,
                              double switch_distance, double detach_distance, double detach_steps, uint8_t* switch_on,
                              int32_t* detach_cnt, double* chosen, hipStream_t s) {
    const int threads = 64;
    hipLaunchKernelGGL(hint_switch_kernel, dim3((B + threads - 1) / threads), dim3(threads), 0, s, B, N, O, V, polygons, valid, states,
                       original, rl_ref, rl_steps, live, switch_distance, detach_distance, detach_steps, switch_on, detach_cnt, chosen);
    return hipGetLastError();
}

}  // namespace mpcgpu
