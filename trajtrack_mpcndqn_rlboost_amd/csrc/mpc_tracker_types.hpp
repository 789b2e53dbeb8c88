// mpc_tracker_types.hpp -- what mpcgpu.hip (host entry points, assembly kernel) and trackgpu.hip (the small per-robot kernels)
// share: the device view of the tracker state and the launchers of trackgpu.hip's kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mpcgpu {

// device view of the tracker state of B robots (all pointers are device memory owned by the caller)
struct TrackerView {
    int B, ref_cap, action_steps;
    double* states;              // [B][3]  in/out
    const double* goals;         // [B][3]
    double* last_actions;        // [B][2]  in/out
    const double* ref;           // [B][ref_cap][3] global reference trajectories, padded
    const int32_t* ref_len;      // [B]
    int32_t* idx_ref;            // [B]     in/out
    const double* stc;           // [B][Nstcobs * 12]
    const double* dyn;           // [B][Ndynobs * 6 * N]
    const double* other;         // [B][3 * N * Nother] or NULL (zeros)
    double* pred_states;         // [B][N][3]  out
    uint8_t* active;             // [B]  in/out
    double tuning[10];           // tuning_params of the work mode
    double base_speed, low_speed, stc_weight, dyn_weight;
};

// launchers of the kernels that live in trackgpu.hip (their own translation unit: the ILP scheduler this file is compiled with
// crashes LLVM's register allocator on the rollout loops)
struct RlLimits { double acc_max, acc_min, angacc_max, angacc_min, speed_min, speed_max, angvel_min, angvel_max; };
hipError_t launch_tracker_window(const TrackerView& t, int N, double* refs_out, hipStream_t s);
hipError_t launch_tracker_apply(const TrackerView& t, int N, double ts, const double* u, double* actions_out, hipStream_t s);
hipError_t launch_rl_reference(int B, const double* agent, int agent_stride, const int64_t* action, double ts, int steps,
                               double ref_speed, const RlLimits& lim, double* rl_ref, hipStream_t s);
hipError_t launch_hint_switch(int B, int N, int O, int V, const double* polygons, const uint8_t* valid, const double* states,
                              const double* original, const double* rl_ref, int rl_steps, const uint8_t* live,
                              double switch_distance, double detach_distance, double detach_steps, uint8_t* switch_on,
                              int32_t* detach_cnt, double* chosen, hipStream_t s);

}  // namespace mpcgpu
