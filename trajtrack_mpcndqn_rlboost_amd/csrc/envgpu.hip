// envgpu.hip -- batched DRL environment step for gfx950 (MI355X): C-ABI of include/mpcgpu_env.h.
//
// One wavefront = one environment, grid = B.  The 64 lanes split the padded obstacle / boundary edges (sector and
// ray distances, inside tests), the obstacles (key-frame poses) and the reference-path segments (projection); the
// cross-lane results are DPP minima and ballots, and one lane finishes the scalar bookkeeping (flags, reward,
// observation vectors).  The map record is read once per step, coalesced (lanes stride over consecutive edges).
// Reference semantics: see the file:line list in include/mpcgpu_env.h; CPU restatement: oracle/rl_env_numpy.py.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <string>

#include "../../include/mpcgpu_env.h"

namespace envgpu {

constexpr int WAVE = 64;
constexpr int NSEG = 8;      // sectors / rays (rays_reward1.py:20)
constexpr int NCORNER = 3;   // corner samples (rays_reward1.py:18)
constexpr int HDR = 16;
constexpr int SDIM = MPCGPU_ENV_STATE_DOUBLES;
constexpr int MAX_OBST = 31;
constexpr double L_SECTOR = 1000.0;  // ext_obsv_sector_and_ray.py:32

struct EnvK {
    mpcgpu_env_params p;
    int rec, o_cum, o_len, o_xy, o_anim, an, o_edge;
};

__host__ __device__ inline int anim_doubles(int K) { return 4 + (K + 1) + 3 * K; }

static bool layout(const mpcgpu_env_params& p, EnvK& k) {
    if (p.n_path_max < 2 || p.n_path_max > WAVE || p.n_obst_max < 0 || p.n_obst_max > MAX_OBST || p.n_kf_max < 1 ||
        p.n_kf_max > 4 || p.n_edge_max < 1)
        return false;
    k.p = p;
    k.o_cum = HDR;
    k.o_len = k.o_cum + p.n_path_max;
    k.o_xy = k.o_len + p.n_path_max;
    k.o_anim = k.o_xy + 2 * p.n_path_max;
    k.an = anim_doubles(p.n_kf_max);
    k.o_edge = k.o_anim + p.n_obst_max * k.an;
    k.rec = k.o_edge + 5 * p.n_edge_max;
    k.rec += k.rec & 1;
    return true;
}

// ---- wave primitives (DPP: cross-lane operands inside VALU instructions) ------------------------------------------
template <int CTRL>
__device__ __forceinline__ double dpp_fill(double x, double fill) {  // lanes without a source receive `fill`
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(__double2loint(fill), lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(__double2hiint(fill), hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_value(double x, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_min(double x) {  // uniform result
    const double inf = INFINITY;
    x = fmin(x, dpp_fill<0x111>(x, inf));  // row_shr:1
    x = fmin(x, dpp_fill<0x112>(x, inf));
    x = fmin(x, dpp_fill<0x114>(x, inf));
    x = fmin(x, dpp_fill<0x118>(x, inf));
    return fmin(fmin(lane_value(x, 15), lane_value(x, 31)), fmin(lane_value(x, 47), lane_value(x, 63)));
}
__device__ __forceinline__ unsigned wave_xor(unsigned m) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m ^= (unsigned)__shfl_xor((int)m, off, WAVE);
    return m;
}
// fmin / fmax as the single instruction they are: the library forms canonicalise every operand the compiler did not compute itself
// (v_max_f64 x, x, x in front of each), three instructions for one in the clip chains of the edge loop.  v_min_f64 / v_max_f64 return
// the other operand for a quiet NaN exactly as fmin / fmax do (IEEE mode): the same values.
__device__ __forceinline__ double min_raw(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double max_raw(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// 1 / x: hardware estimate + one Newton step (~1e-15 relative; the observations are rounded to float32 afterwards)
__device__ __forceinline__ double fast_rcp(double x) {
    const double r = __builtin_amdgcn_rcp(x);
    return r * (2.0 - x * r);
}
__device__ __forceinline__ double normalize_distance(double d) {  // components/utils.py:10-15
    return 2.0 * fast_rcp(1.0 + exp(-0.2 * d)) - 1.0;   // (the observations are rounded to float32)
}

// cos / sin of j * pi / 8, j = 0..15
__constant__ double DIRC[16] = {1.0, 0.92387953251128673848, 0.70710678118654757274, 0.38268343236508978178,
                                0.0, -0.38268343236508978178, -0.70710678118654757274, -0.92387953251128673848,
                                -1.0, -0.92387953251128673848, -0.70710678118654757274, -0.38268343236508978178,
                                0.0, 0.38268343236508978178, 0.70710678118654757274, 0.92387953251128673848};
__constant__ double DIRS[16] = {0.0, 0.38268343236508978178, 0.70710678118654757274, 0.92387953251128673848,
                                1.0, 0.92387953251128673848, 0.70710678118654757274, 0.38268343236508978178,
                                0.0, -0.38268343236508978178, -0.70710678118654757274, -0.92387953251128673848,
                                -1.0, -0.92387953251128673848, -0.70710678118654757274, -0.38268343236508978178};

struct EnvOut {
    float* obs_int; float* obs_ext; double* reward; uint8_t* terminated;
    // in-kernel auto-reset (max_steps > 0): time-limit flag, and where the observation an episode ENDED in is kept
    uint8_t* truncated; float* term_int; float* term_ext; int max_steps;
};

__global__ __launch_bounds__(WAVE, 4) void env_step_kernel(EnvK k, const double* __restrict__ rec_all, double* state_all,
                                                        const int32_t* __restrict__ action, EnvOut out, int B) {
    float* const obs_int = out.obs_int;
    float* const obs_ext = out.obs_ext;
    double* const reward = out.reward;
    uint8_t* const terminated = out.terminated;
    __shared__ double pose[MAX_OBST + 1][4];
    __shared__ __attribute__((aligned(16))) double dirs[16][2];
    const int b = blockIdx.x;
    if (b >= B) return;
    const int lane = threadIdx.x;
    const double* rec = rec_all + (size_t)b * k.rec;
    double* st = state_all + (size_t)b * SDIM;
    const mpcgpu_env_params& P = k.p;
    const int n_path = (int)rec[0], n_obst = (int)rec[1], n_edge = (int)rec[2];
    const double gx = rec[3], gy = rec[4];
    double x = st[0], y = st[1], th = st[2], v = st[3], w = st[4], clock = st[5];
    double last_prog = st[6], steps = st[25];
    int flags = (int)st[7];
    bool act = action != nullptr;
    const double ts = P.time_step;

    // ---- obstacles' clock and the robot (environment.py:199-203, agent.py:97-139); every lane keeps the same copy
    if (act) {
        const int a = action[b];
        clock += ts;
        const int row = a / 3, col = a % 3;
        if (row == 0) v += ts * P.acc_max;
        if (row == 2) v += ts * P.acc_min;
        if (col == 0) w += ts * P.angacc_max;
        if (col == 2) w += ts * P.angacc_min;
        if (v > P.speed_max) v = P.speed_max;
        if (v < P.speed_min) v = P.speed_min;
        if (w > P.angvel_max) w = P.angvel_max;
        if (w < P.angvel_min) w = P.angvel_min;
        th += ts * w;
    }
    // cos / sin of the heading once: the kinematic step and the observation pass use the same angle
    double cth0, sth0;
    sincos(th, &sth0, &cth0);
    if (act) {
        x += ts * v * cth0;
        y += ts * v * sth0;
    }
    // Pass 0 is the step itself.  With in-kernel auto-reset (out.max_steps > 0) an environment whose episode ended is
    // put back to its start state and observed again in pass 1 -- no host round trip, no second launch.
    auto run_pass = [&](const int pass, const double cth, const double sth) -> bool {   // returns true when the episode ended and pass 1 is wanted

    // ---- key-frame pose of obstacle `lane` (obstacle.py:71-88): (x, y, cos rot, sin rot) -> LDS
    if (lane < n_obst) {
        const double* an = rec + k.o_anim + lane * k.an;
        const int kind = (int)an[0], nk = (int)an[2];
        const double* tsv = an + 4;
        const double* kf = an + 4 + (P.n_kf_max + 1);
        const double tm = fmod(clock + an[1], an[3]);
        double px = kf[3 * (nk - 1)], py = kf[3 * (nk - 1) + 1], rot = kf[3 * (nk - 1) + 2];
        double t = 0.0;
        bool found = false;
        for (int i = 0; i < nk; ++i) {
            t += tsv[i];
            if (!found && t <= tm && tm < t + tsv[i + 1]) {
                const double xx = (tm - t) * fast_rcp(tsv[i + 1]);
                const double alpha = kind == 1 ? (1.0 - cos(xx * M_PI)) / 2.0 : xx;
                const double* k0 = kf + 3 * i;
                const double* k1 = kf + 3 * ((i + 1) % nk);
                px = k0[0] * (1.0 - alpha) + k1[0] * alpha;
                py = k0[1] * (1.0 - alpha) + k1[1] * alpha;
                rot = k0[2] * (1.0 - alpha) + k1[2] * alpha;
                found = true;
            }
        }
        double crot, srot;
        sincos(rot, &srot, &crot);
        pose[lane][0] = px; pose[lane][1] = py; pose[lane][2] = crot; pose[lane][3] = srot;
    }
    __syncthreads();

    // ---- the 16 directions theta + j pi/8 (even j = ray / sector centre, odd j = sector borders) live in LDS: every
    //      lane reads the same entry (a broadcast), which keeps 64 VGPRs free for a fourth resident wavefront per SIMD
    if (lane < 16) {
        dirs[lane][0] = cth * DIRC[lane] - sth * DIRS[lane];
        dirs[lane][1] = sth * DIRC[lane] + cth * DIRS[lane];
    }
    __syncthreads();

    // ---- edges: closest point of (edge within sector i), first hit of ray i, crossing parity per outline
    double sec[NSEG], ray[NSEG];
#pragma unroll
    for (int i = 0; i < NSEG; ++i) { sec[i] = INFINITY; ray[i] = INFINITY; }
    unsigned mask = 0u;  // bit 0: padded boundary, bit j+1: obstacle j -- set when the robot is inside that outline
    for (int e = lane; e < n_edge; e += WAVE) {
        const double* ed = rec + k.o_edge + 5 * e;
        double x0 = ed[0], y0 = ed[1], x1 = ed[2], y1 = ed[3];
        const int owner = (int)ed[4];
        if (owner < -1) continue;
        if (owner >= 0) {  // obstacle.py:174-188: position + R * nodes
            const double ox = pose[owner][0], oy = pose[owner][1], c = pose[owner][2], s = pose[owner][3];
            const double a0 = ox + c * x0 - s * y0, b0 = oy + s * x0 + c * y0;
            const double a1 = ox + c * x1 - s * y1, b1 = oy + s * x1 + c * y1;
            x0 = a0; y0 = b0; x1 = a1; y1 = b1;
        }
        const double Px = x0 - x, Py = y0 - y, Qx = x1 - x, Qy = y1 - y;
        const double Ex = Qx - Px, Ey = Qy - Py;
        if ((Py > 0.0) != (Qy > 0.0)) {  // even-odd rule along +x from the robot
            const double xi = Ex * (-Py) * fast_rcp(Ey) + Px;
            if (xi > 0.0) mask ^= 1u << (owner + 1);
        }
        const double ee = Ex * Ex + Ey * Ey, pe = Px * Ex + Py * Ey;
        const double t_free = ee > 0.0 ? -pe * fast_rcp(ee) : 0.0;
        // Line j through the robot with direction d_j meets the edge's carrier at parameter tc; g0 / g1 are the signed
        // distances (x |d| = 1) of P and of the edge direction from that line.  Odd lines are sector borders (each
        // shared by two neighbouring sectors), even lines carry the rays.  The lines
        // are walked in order and only the previous border is kept (registers: residency, see __launch_bounds__).
        struct Line { double g0, g1, tc, dx, dy; };
        asm volatile("" ::: "memory");  // keep the direction reads inside the loop (hoisted, they pin 64 VGPRs again)
        auto line = [&](int j) {
            Line l;
            l.dx = dirs[j][0]; l.dy = dirs[j][1];
            l.g0 = l.dx * Py - l.dy * Px;
            l.g1 = l.dx * Ey - l.dy * Ex;
            l.tc = -l.g0 * fast_rcp(l.g1);   // inf / nan when the edge is parallel to the line: handled below
            return l;
        };
        // Line j + 8 is line j walked the other way: g0 and g1 change sign, tc stays.  Sector i and the sector opposite to it (i + 4)
        // share their two border carriers, ray i and ray i + 4 their carrier: nine line evaluations per edge instead of sixteen.
        auto sector = [&](int i, const Line& lo, const Line& up, double sgn) {
            // wedge = {cross(d_lower, X) >= 0} and {cross(d_upper, X) <= 0}: clip the edge's parameter range [0, 1]
            const double lg1 = sgn * lo.g1, lg0 = sgn * lo.g0, ug1 = sgn * up.g1, ug0 = sgn * up.g0;
            double t0 = 0.0, t1 = 1.0;
            bool empty = false;
            if (lg1 > 0.0) t0 = max_raw(t0, lo.tc);
            else if (lg1 < 0.0) t1 = min_raw(t1, lo.tc);
            else if (lg0 < 0.0) empty = true;
            if (ug1 < 0.0) t0 = max_raw(t0, up.tc);
            else if (ug1 > 0.0) t1 = min_raw(t1, up.tc);
            else if (ug0 > 0.0) empty = true;
            if (!empty && t0 <= t1) {
                const double tt = min_raw(max_raw(t_free, t0), t1);
                const double cx = Px + tt * Ex, cy = Py + tt * Ey;
                sec[i] = min_raw(sec[i], cx * cx + cy * cy);  // squared; the root is taken once, after the reduction
            }
        };
        auto rayhit = [&](int i, const Line& ce, double sgn) {
            // ray i: P + t E = s d with t = tc of the centre line; s follows from the projection on d (|d| = 1)
            if (ce.g1 != 0.0) {
                const double t = ce.tc;
                const double sd = sgn * ((Px + t * Ex) * ce.dx + (Py + t * Ey) * ce.dy);
                if (sd >= 0.0 && t >= 0.0 && t <= 1.0 && sd <= L_SECTOR) ray[i] = min_raw(ray[i], sd);
            }
        };
        Line lower = line(7);   // line 15 = line 7 reversed: the lower border of sector 0
#pragma unroll
        for (int i = 0; i < NSEG / 2; ++i) {
            const Line centre = line(2 * i);
            const Line upper = line(2 * i + 1);
            // sector i lies between lines 2i - 1 and 2i + 1; for i = 0 the lower border is line 15 = -line 7
            {   // sector i
                Line lo = lower;
                if (i == 0) { lo.g0 = -lo.g0; lo.g1 = -lo.g1; }
                sector(i, lo, upper, 1.0);
                // the opposite sector i + 4: borders 2i + 7 = -(2i - 1) ... and 2i + 9 = -(2i + 1)
                sector(i + NSEG / 2, lo, upper, -1.0);
            }
            rayhit(i, centre, 1.0);
            rayhit(i + NSEG / 2, centre, -1.0);
            lower = upper;
        }
    }
    mask = wave_xor(mask);
    const bool in_obstacle = (mask & ~1u) != 0u;
    // 16 minima over 64 lanes as ONE butterfly: every exchange halves the number of values a lane still carries
    // (8 + 4 + 2 + 1 + 1 + 1 = 17 exchanges instead of 16 x 6).  Afterwards lane l holds value number
    // 8 b5 + 4 b4 + 2 b3 + b2 (b_k = bit k of l); values 0..7 = squared sector distances, 8..15 = ray distances.
    double red;
    {
        double v8[8], v4[4], v2[2];
        const bool h5 = lane & 32, h4 = lane & 16, h3 = lane & 8, h2 = lane & 4;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const double keep = h5 ? ray[i] : sec[i], send = h5 ? sec[i] : ray[i];
            v8[i] = min_raw(keep, __shfl_xor(send, 32, WAVE));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const double keep = h4 ? v8[4 + i] : v8[i], send = h4 ? v8[i] : v8[4 + i];
            v4[i] = min_raw(keep, __shfl_xor(send, 16, WAVE));
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const double keep = h3 ? v4[2 + i] : v4[i], send = h3 ? v4[i] : v4[2 + i];
            v2[i] = min_raw(keep, __shfl_xor(send, 8, WAVE));
        }
        const double keep = h2 ? v2[1] : v2[0], send = h2 ? v2[0] : v2[1];
        red = min_raw(keep, __shfl_xor(send, 4, WAVE));
        red = min_raw(red, __shfl_xor(red, 2, WAVE));
        red = min_raw(red, __shfl_xor(red, 1, WAVE));
    }
    // ---- external observation with one-step memory (ext_obsv_sector_and_ray.py:66-74): 16 lanes, one entry each
    if ((lane & 3) == 0) {
        const int idx = ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
        // inside a padded obstacle the robot itself belongs to sector ∩ obstacle: every distance is 0
        const double d = in_obstacle ? 0.0 : (idx < NSEG ? sqrt(red) : red);
        const float o = (float)normalize_distance(d);
        float* oe = obs_ext + (size_t)b * MPCGPU_ENV_EXTERNAL_OBS;
        oe[idx] = o;
        oe[2 * NSEG + idx] = (float)st[8 + idx];
        st[8 + idx] = (double)o;
    }

    // ---- path progress = LineString.project: first closest segment (environment.py:118)
    const double* cum = rec + k.o_cum;
    const double* len = rec + k.o_len;
    const double* pxy = rec + k.o_xy;
    double dist = INFINITY, tproj = 0.0, cpx = 0.0, cpy = 0.0;
    if (lane < n_path - 1) {
        const double ax = pxy[2 * lane], ay = pxy[2 * lane + 1];
        const double ex = pxy[2 * lane + 2] - ax, ey = pxy[2 * lane + 3] - ay;
        const double den = ex * ex + ey * ey;
        double t = den == 0.0 ? 0.0 : ((x - ax) * ex + (y - ay) * ey) * fast_rcp(den);
        t = fmin(1.0, fmax(0.0, t));
        cpx = ax + t * ex; cpy = ay + t * ey;
        tproj = t;
        dist = sqrt((x - cpx) * (x - cpx) + (y - cpy) * (y - cpy));
    }
    const double cte = wave_min(dist);
    const int iseg = __ffsll((long long)__ballot(dist == cte)) - 1;
    const double progress = cum[iseg] + lane_value(tproj, iseg) * len[iseg];
    cpx = lane_value(cpx, iseg); cpy = lane_value(cpy, iseg);
    // sample point at progress + offset (LineString.interpolate, clamped to the line)
    double spx = cpx, spy = cpy;
    if (P.sample_offset != 0.0) {
        const double s = progress + P.sample_offset;
        const double total = cum[n_path - 1];
        if (s <= 0.0) { spx = pxy[0]; spy = pxy[1]; }
        else if (s >= total) { spx = pxy[2 * (n_path - 1)]; spy = pxy[2 * (n_path - 1) + 1]; }
        else {
            const unsigned long long bal = __ballot(lane < n_path - 1 && s < cum[lane < n_path - 1 ? lane + 1 : 0]);
            const int i = bal ? __ffsll((long long)bal) - 1 : n_path - 2;
            const double t = (s - cum[i]) * fast_rcp(len[i]);
            spx = pxy[2 * i] + t * (pxy[2 * i + 2] - pxy[2 * i]);
            spy = pxy[2 * i + 1] + t * (pxy[2 * i + 3] - pxy[2 * i + 1]);
        }
    }
    // first node whose cumulative length reaches the progress (int_obsv_reference_path_corner.py:29-33)
    const unsigned long long reach = __ballot(lane < n_path && cum[lane < n_path ? lane : 0] >= progress);
    int icorner = reach ? __ffsll((long long)reach) - 1 : n_path - 1;

    // ---- path observations (components/int_obsv_reference_path_{sample,corner}.py): lanes 0..3, one point each
    float* oi = obs_int + (size_t)b * MPCGPU_ENV_INTERNAL_OBS;
    if (lane < 1 + NCORNER) {
        double qx = spx, qy = spy;
        if (lane > 0) {
            int ic = icorner + lane - 1;
            if (ic > n_path - 1) ic = n_path - 1;
            qx = pxy[2 * ic]; qy = pxy[2 * ic + 1];
        }
        // cos / sin of (bearing - theta) and the normalised distance; atan2(0, 0) = 0 for a coincident point
        const double ddx = qx - x, ddy = qy - y;
        const double d = sqrt(ddx * ddx + ddy * ddy);
        double cr = cth, sr = -sth;
        if (d > 0.0) { const double id = fast_rcp(d); cr = (ddx * cth + ddy * sth) * id; sr = (ddy * cth - ddx * sth) * id; }
        float* o = oi + 2 + 3 * lane;
        o[0] = (float)cr; o[1] = (float)sr; o[2] = (float)normalize_distance(d);
    }
    // ---- status flags, sticky (environment.py:113-116); uniform, every lane keeps a copy
    if (in_obstacle) flags |= 1;
    if (!(mask & 1u)) flags |= 2;
    if (sqrt((gx - x) * (gx - x) + (gy - y) * (gy - y)) < P.radius) flags |= 4;
    const bool collided = (flags & 3) != 0, reached = (flags & 4) != 0;
    if (act) { steps += 1.0; }
    const bool timeout = out.max_steps > 0 && steps >= (double)out.max_steps && !(collided || reached);

    if (lane == 0) {
        // ---- speed and angular velocity (components/int_obsv_speed.py, int_obsv_angular_velocity.py)
        oi[0] = (float)(2.0 * (v - P.speed_min) * fast_rcp(P.speed_max - P.speed_min) - 1.0);
        // the reference normalises the angular velocity with the angular ACCELERATION limits (int_obsv_angular_velocity.py:13-19)
        oi[1] = (float)(2.0 * (w - P.angacc_min) * fast_rcp(P.angacc_max - P.angacc_min) - 1.0);
        if (pass == 0) {
            // ---- reward R1 (rays_reward1.py:26-39; summed in component order)
            if (act) {
                double r = collided ? -P.collision_factor : 0.0;
                r += -ts * P.cross_track_factor * cte * cte;
                r += reached ? P.reach_goal_factor : 0.0;
                const double err = copysign(1.0, P.reference_speed) * (v - P.reference_speed);
                r += -ts * P.excessive_speed_factor * fmax(0.0, err);
                r += P.path_progress_factor * (progress - last_prog);
                if (reward) reward[b] = r;
            } else if (reward) {
                reward[b] = 0.0;
            }
            if (terminated) terminated[b] = (collided || reached) ? 1 : 0;
            if (out.truncated) out.truncated[b] = timeout ? 1 : 0;
            st[26] = (double)flags;   // flags of this step, still readable after an in-kernel reset
        }
        st[0] = x; st[1] = y; st[2] = th; st[3] = v; st[4] = w; st[5] = clock;
        st[6] = (act || pass == 1) ? (pass == 1 ? 0.0 : progress) : last_prog;
        st[7] = (double)flags;
        st[24] = progress;
        st[25] = steps;
    }
    return pass == 0 && out.max_steps > 0 && act && (collided || reached || timeout);
    };  // run_pass

    if (!run_pass(0, cth0, sth0)) return;

    // ---- the episode ended: keep its last observation, go back to the start state (environment.py:166-186) and
    //      observe once more.  The observation memory (st[8..23]) is not cleared -- the reference's component keeps it.
    __threadfence();
    __syncthreads();
    if (out.term_int && lane < MPCGPU_ENV_INTERNAL_OBS)
        out.term_int[(size_t)b * MPCGPU_ENV_INTERNAL_OBS + lane] = obs_int[(size_t)b * MPCGPU_ENV_INTERNAL_OBS + lane];
    if (out.term_ext && lane < MPCGPU_ENV_EXTERNAL_OBS)
        out.term_ext[(size_t)b * MPCGPU_ENV_EXTERNAL_OBS + lane] = obs_ext[(size_t)b * MPCGPU_ENV_EXTERNAL_OBS + lane];
    __syncthreads();
    x = rec[5]; y = rec[6]; th = rec[7]; v = rec[8]; w = rec[9];
    clock = 0.0; flags = 0; steps = 0.0; last_prog = 0.0; act = false;
    sincos(th, &sth0, &cth0);
    run_pass(1, cth0, sth0);
}

thread_local std::string g_err;
static int fail(const char* what, hipError_t e = hipSuccess) {
    char buf[256];
    if (e != hipSuccess) snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    else snprintf(buf, sizeof buf, "%s", what);
    g_err = buf;
    return -1;
}

}  // namespace envgpu

extern "C" {

int32_t mpcgpu_env_record_doubles(const mpcgpu_env_params* params) {
    envgpu::EnvK k;
    if (!params || !envgpu::layout(*params, k)) return envgpu::fail("invalid mpcgpu_env_params (P 2..64, M 0..31, K 1..4, E >= 1)");
    return k.rec;
}

static int32_t launch_env(int32_t device, const mpcgpu_env_params* params, int32_t B, const double* records, double* state,
                          const int32_t* action, envgpu::EnvOut out, void* stream) {
    using namespace envgpu;
    EnvK k;
    if (!params || !layout(*params, k)) return fail("invalid mpcgpu_env_params (P 2..64, M 0..31, K 1..4, E >= 1)");
    if (params->num_segments != NSEG || params->corner_samples != NCORNER)
        return fail("only num_segments = 8 and corner_samples = 3 are built (rays_reward1.py:18-20)");
    if (B < 0 || !records || !state || !out.obs_int || !out.obs_ext) return fail("null pointer / negative batch");
    if (B == 0) return 0;
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return fail("hipSetDevice", e);
    hipLaunchKernelGGL(env_step_kernel, dim3(B), dim3(WAVE), 0, (hipStream_t)stream, k, records, state, action, out, (int)B);
    e = hipGetLastError();
    if (e != hipSuccess) return fail("env_step_kernel launch", e);
    return 0;
}

int32_t mpcgpu_env_step_dev(int32_t device, const mpcgpu_env_params* params, int32_t B, const double* records,
                            double* state, const int32_t* action, float* obs_internal, float* obs_external,
                            double* reward, uint8_t* terminated, void* stream) {
    envgpu::EnvOut out{obs_internal, obs_external, reward, terminated, nullptr, nullptr, nullptr, 0};
    return launch_env(device, params, B, records, state, action, out, stream);
}

int32_t mpcgpu_env_step_autoreset_dev(int32_t device, const mpcgpu_env_params* params, int32_t B, const double* records,
                                      double* state, const int32_t* action, float* obs_internal, float* obs_external,
                                      double* reward, uint8_t* terminated, uint8_t* truncated, float* terminal_obs_internal,
                                      float* terminal_obs_external, int32_t max_episode_steps, void* stream) {
    if (!action) return envgpu::fail("auto-reset needs actions (use mpcgpu_env_step_dev to observe)");
    if (max_episode_steps <= 0) return envgpu::fail("max_episode_steps must be positive");
    envgpu::EnvOut out{obs_internal, obs_external, reward, terminated, truncated, terminal_obs_internal,
                       terminal_obs_external, max_episode_steps};
    return launch_env(device, params, B, records, state, action, out, stream);
}

const char* mpcgpu_env_last_error(void) { return envgpu::g_err.c_str(); }

}  // extern "C"
