/*
 * mpcgpu_env.h -- C-ABI of the batched DRL environment step in libmpcgpu.so.
 *
 * Replaces, for B independent environments per launch (paths relative to /root/reference/src/pkg_dqn/environment/):
 *
 *     env.step(action)                       environment.py:199-213   (obstacles, robot, status, observation, reward)
 *     env.update_status(); env.get_observation()    environment.py:96-122,156-161  (observe-only: src/main.py:181-189)
 *     MobileRobot.step(action_index, ts)     agent.py:97-139
 *     Obstacle.step / Animation.get_keyframe obstacle.py:71-88,125-127
 *     SectorAndRayObservation.external_obs   components/ext_obsv_sector_and_ray.py:31-81
 *     Speed / AngularVelocity / ReferencePathSample / ReferencePathCorner observations   components/int_obsv_*.py
 *     Collision / CrossTrack / ReachGoal / ExcessiveSpeed / PathProgress rewards          components/reward_*.py
 *       as assembled by TrajectoryPlannerEnvironmentRaysReward1                           variants/rays_reward1.py:26-39
 *
 * Maps (padded outlines, key frames, reference path) are INPUTS, packed once per environment into a record of doubles
 * (layout below; the Python side, rl_env.py, builds it).  All pointers are DEVICE pointers; the call enqueues one kernel
 * on `stream` and returns (no synchronisation).  Plain pointers and sizes, no ownership taken, no exceptions:
 * 0 = ok, < 0 = error (text via mpcgpu_env_last_error, thread-local).  There is no CPU fallback.
 *
 * Record of environment b (doubles), R = mpcgpu_env_record_doubles(params):
 *     [0] n_path  [1] n_obstacles  [2] n_edges  [3] goal_x  [4] goal_y  [5..9] start state x, y, theta, v, w (used by the
 *     in-kernel auto-reset)  [10..15] reserved
 *     path_cum [P]      cumulative length at node i (sequential float64 sum, path_cum[i+1] = path_cum[i] + path_len[i])
 *     path_len [P]      length of segment i -> i+1
 *     path_xy  [P][2]
 *     anim     [M][4 + (K+1) + 3K]   kind (0 linear, 1 cosine), offset, n_keyframes, cycle length,
 *                                    time_steps[K+1], keyframes[K][x, y, rotation]        (obstacle.py:52-105)
 *     edges    [E][5]   x0, y0, x1, y1 in the owner's body frame, owner (-1: padded boundary, world frame;
 *                       j >= 0: padded outline of obstacle j; -2: unused row)
 * with P, M, K, E = n_path_max, n_obst_max, n_kf_max, n_edge_max.
 *
 * State of environment b (32 doubles, read and written in place):
 *     [0..4] x, y, theta, v, w   [5] obstacle clock   [6] last path progress (reward_path_progress.py)
 *     [7] flags: 1 collided with obstacle | 2 collided with boundary | 4 reached goal  (sticky, environment.py:113-116)
 *     [8..23] first 16 entries of the previous external observation (the observation's one-step memory)
 *     [24] path progress (output)   [25] step counter   [26] flags of the last step (kept across an in-kernel reset)
 *     [27..31] reserved
 */
#ifndef MPCGPU_ENV_H
#define MPCGPU_ENV_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPCGPU_ENV_STATE_DOUBLES 32
#define MPCGPU_ENV_INTERNAL_OBS 14  /* speed, angular velocity, 1 path sample x 3, 3 corners x 3 */
#define MPCGPU_ENV_EXTERNAL_OBS 32  /* 8 sectors, 8 rays, and both from the previous step */

typedef struct mpcgpu_env_params {
    int32_t n_path_max;   /* P: 2..64 */
    int32_t n_obst_max;   /* M: 0..31 */
    int32_t n_kf_max;     /* K: 1..4  */
    int32_t n_edge_max;   /* E */
    int32_t num_segments; /* 8  (rays_reward1.py:20)  -- the only value built */
    int32_t corner_samples; /* 3 (rays_reward1.py:18) -- the only value built */
    double time_step;               /* 0.2 */
    double sample_offset;           /* reference_path_sample_offset, 0 */
    double collision_factor;        /* 4 */
    double reach_goal_factor;       /* 3 */
    double cross_track_factor;      /* 0.05 */
    double excessive_speed_factor;  /* 2 * path_progress_factor = 4 */
    double reference_speed;         /* SPEED_MAX * 0.8 = 1.2 */
    double path_progress_factor;    /* 2 */
    /* MobileRobotSpecification, agent.py:7-16 */
    double radius, speed_min, speed_max, angvel_min, angvel_max, acc_min, acc_max, angacc_min, angacc_max;
} mpcgpu_env_params;

/* doubles per environment record for these maxima (the layout above), < 0 on invalid params */
int32_t mpcgpu_env_record_doubles(const mpcgpu_env_params* params);

/*
 * One environment step for B environments (replaces env.step, environment.py:199-213).
 *   records [B x R]   state [B x 32]   action [B] in 0..8 (agent.py:104-118), or NULL = observe only
 *   obs_internal [B x 14] float   obs_external [B x 32] float   reward [B] (may be NULL)   terminated [B] (may be NULL)
 */
int32_t mpcgpu_env_step_dev(int32_t device, const mpcgpu_env_params* params, int32_t B, const double* records,
                            double* state, const int32_t* action, float* obs_internal, float* obs_external,
                            double* reward, uint8_t* terminated, void* stream);

/*
 * The same step with the episode bookkeeping of a vectorised environment done INSIDE the kernel (what SB3's VecEnv +
 * gym's TimeLimit do around env.step for the reference's training, src/test_block_rl.py:68-69, environment/__init__.py:
 * 15-25): an environment that terminated (collision / goal) or reached max_episode_steps is put back to the start
 * state of its record, observed again, and that observation is what obs_internal / obs_external hold on return; the
 * observation the episode ended in goes to terminal_obs_* (rows of environments that did not end are left untouched).
 *   truncated [B]: 1 = ended by the step limit (not a terminal state for the bootstrap).  terminal_obs_* may be NULL.
 */
int32_t mpcgpu_env_step_autoreset_dev(int32_t device, const mpcgpu_env_params* params, int32_t B, const double* records,
                                      double* state, const int32_t* action, float* obs_internal, float* obs_external,
                                      double* reward, uint8_t* terminated, uint8_t* truncated, float* terminal_obs_internal,
                                      float* terminal_obs_external, int32_t max_episode_steps, void* stream);

const char* mpcgpu_env_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* MPCGPU_ENV_H */
