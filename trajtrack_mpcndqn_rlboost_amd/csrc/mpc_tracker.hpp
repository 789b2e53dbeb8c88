// mpc_tracker.hpp -- the batched tracker harness on the device and the reference generator of the
// hybrid DQN -> MPC loop (row f2).  One control tick of B robots without the padded 21 KB parameter vector and without a host
// round trip:
//   tracker_window_kernel    get_local_ref_traj (src/mpc_traj_tracker/trajectory_generator.py:206-232): nearest sample of the
//                            global reference inside the window [idx - a, idx + 5a), then N rows from there, tail padded;
//   tracker_assemble_kernel  check_termination_condition (:156-162) + run_step's parameter assembly (:251-275) written DIRECTLY as
//                            the compact workspace record of the solve kernel: prep_problem (mpc_kernels.hpp) reads the same
//                            parameter indices through a source that maps them onto the tracker's own arrays -- the record is
//                            bitwise what prep_kernel makes of the assembled vector;
//   tracker_apply_kernel     the post-solve part of run_step (:325-339): take action_steps inputs, roll the prediction out;
//   rl_reference_kernel      the DQN's proposal: decoded acceleration pair + 20-step rollout with decaying turn rate
//                            (src/pkg_dqn/environment/agent.py:86-145, src/main.py:193-202);
//   hint_switch_kernel       HintSwitcher.switch (src/main_pre.py:27-52) for every robot + the reference it tracks this tick
//                            (ref_traj_filter with decay 1 = the proposal with the original heading column, src/main.py:34-41).
#pragma once
#include "mpc_kernels.hpp"
#include "mpc_tracker_types.hpp"

namespace mpcgpu {

// ---- parameter source of prep_problem: index of the reference's parameter vector (mpc_generator.py:179-188) -> tracker arrays
struct TrackerParams {
    const double* state; const double* last_u; const double* refs;   // this robot's rows
    const double* stc; const double* dyn; const double* other;
    const double* tuning;
    double vref, stc_w, dyn_w;
    int N, r0, c0, os0, od0, qs0, qd0;
    __device__ __forceinline__ double operator[](int i) const {
        if (i >= od0) {
            if (i < qs0) return dyn[i - od0];
            return i < qd0 ? stc_w : dyn_w;
        }
        if (i >= os0) return stc[i - os0];
        if (i >= c0) return other ? other[i - c0] : 0.0;
        if (i >= r0 + 3 * N) return vref;
        if (i >= r0) return refs[i - r0];
        if (i >= 8) return tuning[i - 8];
        if (i >= 6) return last_u[i - 6];
        if (i >= 3) return refs[3 * (N - 1) + i - 3];      // finish state = last row of the reference that is tracked
        return state[i];
    }
};

// one wavefront per robot: termination test, speed rule, compact record.  refs [B][N][3]: the reference every robot tracks
__global__ __launch_bounds__(WAVE) void tracker_assemble_kernel(KParams kp, TrackerView t, const double* __restrict__ refs,
                                                                double* __restrict__ ws, int* counts) {
    const int b = blockIdx.x, lane = threadIdx.x;
    if (b >= t.B) return;
    const int N = kp.N;
    const double* st = t.states + 3 * b;
    const double* gl = t.goals + 3 * b;
    // check_termination_condition: within 5 cm of the goal in x and y with a last speed below 0.05
    if (lane == 0) {
        const bool near = fabs(st[0] - gl[0]) <= 0.05 && fabs(st[1] - gl[1]) <= 0.05;
        if (near && fabs(t.last_actions[2 * b]) < 0.05) t.active[b] = 0;
    }
    TrackerParams p;
    p.state = st; p.last_u = t.last_actions + 2 * b; p.refs = refs + (size_t)b * N * 3;
    p.stc = t.stc + (size_t)b * kp.Nstcobs * STCW; p.dyn = t.dyn + (size_t)b * kp.Ndynobs * 6 * N;
    p.other = t.other ? t.other + (size_t)b * 3 * N * kp.Nother : nullptr;
    p.tuning = t.tuning;
    // speed reference: constant, scaled down near the FINAL goal, floored by low_speed (trajectory_generator.py:257-264)
    const double dist = hypot(st[0] - gl[0], st[1] - gl[1]);
    p.vref = dist >= t.base_speed * N * kp.ts ? t.base_speed : fmax(dist / N / kp.ts, t.low_speed);
    p.stc_w = t.stc_weight; p.dyn_w = t.dyn_weight;
    p.N = N; p.r0 = kp.r0; p.c0 = kp.c0; p.os0 = kp.os0; p.od0 = kp.od0; p.qs0 = kp.qs0; p.qd0 = kp.qd0;
    prep_problem(kp, p, ws + (size_t)b * kp.ws_stride, counts, lane);
}

}  // namespace mpcgpu
