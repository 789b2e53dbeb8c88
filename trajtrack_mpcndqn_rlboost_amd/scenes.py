"""Seeded synthetic scenes -> parameter vectors ``p`` for batches of independent robots.

The layout of ``p`` is the reference's (``src/mpc_traj_tracker/mpc/mpc_generator.py:179-188``, assembled per
robot at ``src/mpc_traj_tracker/trajectory_generator.py:272-275``).  Scene content follows the reference's own
demo setting: corridor walls of scene 1 (``src/pkg_dqn/utils/map.py:294-297``) and one box on the path, all
inflated by 0.8 m (``src/main.py:110``); dynamic obstacles are discs of radius 0.8+0.8 m with constant-velocity
prediction ``pos + (k+1)*delta`` (``src/main.py:31,77-85``); reference points 0.24 m apart
(= lin_vel_max*high_speed*ts, ``trajectory_generator.py:133-134``); tuning weights of the 'work' mode
(``trajectory_generator.py:129-134``); obstacle weights 1e3 (``trajectory_generator.py:59``).
No reference code is needed to generate them.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np

from .config import MpcConfig

WALLS = [((0.0, 9.0), (1.5, 1.6)), ((0.0, 9.0), (8.4, 8.5)), ((11.0, 16.0), (1.5, 1.6)), ((11.0, 16.0), (8.4, 8.5))]
INFLATE = 0.8
DYN_OBS_SIZE = 0.8 + 0.8


def rect_halfspaces(x0, x1, y0, y1):
    """(b, a0, a1) of axis-aligned boxes, normalised like the reference's facet enumeration
    (``src/util/utils_geo.py:33-59``: A (v - centre) = 1 on every facet, b = A centre + 1).
    Inputs broadcast; output [..., 12] = b[4] + a0[4] + a1[4]."""
    x0, x1, y0, y1 = np.broadcast_arrays(*[np.asarray(a, dtype=float) for a in (x0, x1, y0, y1)])
    cx, cy, hx, hy = 0.5 * (x0 + x1), 0.5 * (y0 + y1), 0.5 * (x1 - x0), 0.5 * (y1 - y0)
    a0 = np.stack([1.0 / hx, -1.0 / hx, np.zeros_like(hx), np.zeros_like(hx)], axis=-1)
    a1 = np.stack([np.zeros_like(hy), np.zeros_like(hy), 1.0 / hy, -1.0 / hy], axis=-1)
    b = a0 * cx[..., None] + a1 * cy[..., None] + 1.0
    return np.concatenate([b, a0, a1], axis=-1)


def work_mode_weights(cfg: MpcConfig) -> np.ndarray:
    """tuning_params of every non-'aligning' mode (trajectory_generator.py:129-130)."""
    return np.array([cfg.qpos, cfg.qvel, cfg.qtheta, cfg.lin_vel_penalty, cfg.ang_vel_penalty,
                     cfg.qpN, cfg.qthetaN, cfg.qrpd, cfg.lin_acc_penalty, cfg.ang_acc_penalty], dtype=float)


def make_batch(cfg: MpcConfig, B: int, n_dyn: int = 8, seed: int = 1234, n_other: int = 0,
               with_box: bool = True, with_walls: bool = True, v_init_range=(0.0, 1.2),
               dyn_clearance: Optional[float] = None, box_clearance: Optional[float] = None,
               on_track: bool = False, n_block=None, block_overlap=(0.1, 0.6), block_first_step: int = 8,
               box_overlap=None, box_overlap_share: float = 1.0, dyn_weight=1e3,
               out: Optional[np.ndarray] = None) -> Dict[str, np.ndarray]:
    """Returns dict(p=[B, np] float64, start=[B,3], ref=[B,N,3]).

    ``v_init_range``: range of the previously applied linear speed (p[6]).  Close to the reference speed
    (1.2 m/s) the acceleration constraints stay inactive and the ALM loop converges in two outer iterations.
    ``dyn_clearance``: None = the benchmark family (discs anywhere within 3 m of the path, crossing it: 8 discs of
    radius 1.6 m cover more area than the corridor has, most problems have no collision-free plan and the solver
    stops at its iteration caps).  A number = the "passing" family: every disc keeps that gap (metres) between its
    hard radius and the reference path over the whole horizon (it travels parallel to the path on either side), so
    a collision-free plan exists, the soft margin (social_margin) still bends the plan, and the solver converges.
    ``box_clearance``: None = the box sits on the path (benchmark family: the product-of-squared-hinges constraint is
    flat at the box edge, so the penalty method leaves ||F2|| > 1e-4 at the iteration caps).  A number = the inflated
    box stands beside the path with that gap (metres); it is still evaluated at every step of every iteration.
    ``on_track``: the robot starts aligned with a straight reference (what a tick in the middle of a closed-loop run
    looks like) instead of up to 0.45 rad off a reference with a corner.
    ``n_block``: None, or (lo, hi) = the "avoidance" family: that many of the discs (drawn per problem) stand ON the
    reference path -- their hard ellipse (radius 1.6 m, ``src/main.py:31,77-85``) covers the path by ``block_overlap``
    metres at a step >= ``block_first_step`` and they drift along the path, always on the side of the nearer corridor wall,
    so the detour leads towards the middle of the corridor (>= 2 m of free width) and the hard constraint
    (``mpc_generator.py:229-241,272``) is active at the optimum.  The other discs follow ``dyn_clearance``.
    ``box_overlap``: None, or (lo, hi) = the inflated box covers the path by that many metres (same side rule) in a share
    ``box_overlap_share`` of the problems (the product-of-squared-hinges constraint is flat at the box edge: few of those
    converge); in the others it stands beside the path as with ``box_clearance``.
    ``dyn_weight``: the soft-term weights q_dyn the tracker passes (``set_obstacle_weights(dyn_weights=...)``,
    ``trajectory_generator.py:59,96-113``; default 1e3), a number or (lo, hi) = log-uniform per problem.  With the default
    the soft margin (weight 1e3 on an ellipse 0.2 m wider) keeps the plan OUTSIDE the hard ellipse; with a weight below
    ~50 the path-deviation cost wins and the plan rests ON the hard ellipse (active constraint, positive multiplier).
    ``out``: a C-contiguous float64 [B, np] array to fill instead of allocating one."""
    N = int(cfg.N_hor)
    off = cfg.offsets()
    assert n_dyn <= cfg.Ndynobs and n_other <= cfg.Nother
    rng = np.random.default_rng(seed)
    if out is None:
        p = np.zeros((B, cfg.num_params))
    else:       # a caller-owned buffer (bench.py: one pinned allocation for every family -- first-touch page faults of a fresh
        p = out  # 2.8 GB array cost more than generating its content)
        assert p.shape == (B, cfg.num_params) and p.dtype == np.float64 and p.flags["C_CONTIGUOUS"]
        p[...] = 0.0
    step = cfg.lin_vel_max * cfg.high_speed * cfg.ts  # 0.24 m

    # ---- start pose and reference polyline (straight, one corner)
    x = rng.uniform(0.0, 2.0, B)
    y = rng.uniform(3.0, 7.0, B)
    th = rng.uniform(-0.3, 0.3, B)
    head0 = rng.uniform(-0.15, 0.15, B)
    turn = rng.uniform(-0.6, 0.6, B) * (rng.random(B) < 0.7)
    corner = rng.integers(6, N, B)                       # reference index where the path bends
    if on_track:
        turn = 0.0 * turn
        th = head0 + 0.1 * th
    ks = np.arange(N)[None, :]
    before = np.minimum(ks + 1, corner[:, None])
    after = np.maximum(ks + 1 - corner[:, None], 0)

    def polyline(turn_):
        h1 = head0 + turn_
        px = x[:, None] + step * (before * np.cos(head0)[:, None] + after * np.cos(h1)[:, None])
        py = y[:, None] + step * (before * np.sin(head0)[:, None] + after * np.sin(h1)[:, None])
        return px, py, h1
    rx_, ry_, h1 = polyline(turn)
    bad = (ry_[:, -1] < 3.0) | (ry_[:, -1] > 7.0)
    turn = np.where(bad, -turn, turn)
    rx_, ry_, h1 = polyline(turn)
    bad = (ry_[:, -1] < 3.0) | (ry_[:, -1] > 7.0)
    turn = np.where(bad, 0.0, turn)
    rx_, ry_, h1 = polyline(turn)
    rth = np.where(ks + 1 <= corner[:, None], head0[:, None], h1[:, None])
    ref = np.stack([rx_, ry_, rth], axis=-1)             # [B, N, 3]

    p[:, 0] = x; p[:, 1] = y; p[:, 2] = th
    p[:, 3:6] = ref[:, -1, :]                             # finish_state = current_ref_traj[-1] (:254)
    p[:, 6] = rng.uniform(v_init_range[0], v_init_range[1], B)  # last_u
    p[:, 7] = rng.uniform(-0.2, 0.2, B)
    p[:, off["q"]:off["q"] + 10] = work_mode_weights(cfg)
    p[:, off["r"]:off["r"] + 3 * N] = ref.reshape(B, 3 * N)
    p[:, off["vref"]:off["vref"] + N] = cfg.lin_vel_max * cfg.high_speed

    # ---- other robots: constant-velocity predictions near the path
    for j in range(n_other):
        k0 = rng.integers(2, N, B)
        cx = ref[np.arange(B), k0, 0] + rng.uniform(-0.6, 0.6, B)
        cy = ref[np.arange(B), k0, 1] + rng.uniform(-0.6, 0.6, B)
        vx, vy = rng.uniform(-0.1, 0.1, B), rng.uniform(-0.1, 0.1, B)
        blk = np.zeros((B, N, 3))
        blk[:, :, 0] = cx[:, None] + vx[:, None] * (ks - k0[:, None])
        blk[:, :, 1] = cy[:, None] + vy[:, None] * (ks - k0[:, None])
        p[:, off["c"] + j * 3 * N: off["c"] + (j + 1) * 3 * N] = blk.reshape(B, 3 * N)

    # ---- static obstacles: corridor walls + one box ahead on the path, inflated
    o = 0
    if with_walls:
        for (wx, wy) in WALLS:
            p[:, off["os"] + 12 * o: off["os"] + 12 * (o + 1)] = rect_halfspaces(
                wx[0] - INFLATE, wx[1] + INFLATE, wy[0] - INFLATE, wy[1] + INFLATE)
            o += 1
    if with_box:
        kb = rng.integers(min(12, N - 1), min(18, N - 1) + 1, B)   # 3.1 .. 4.6 m ahead
        lat = rng.uniform(-0.5, 0.5, B)
        bx = ref[np.arange(B), kb, 0] - lat * np.sin(ref[np.arange(B), kb, 2])
        by = ref[np.arange(B), kb, 1] + lat * np.cos(ref[np.arange(B), kb, 2])
        hx = 0.5 * rng.uniform(1.0, 2.0, B) + INFLATE
        hy = 0.5 * rng.uniform(1.0, 2.0, B) + INFLATE
        if box_overlap is not None:     # partly ON the path: the nearer edge lies that far beyond the path (axis-aligned box, path ~ along x)
            yb = ref[np.arange(B), kb, 1]
            side = np.where(yb >= 5.0, 1.0, -1.0)
            on = rng.random(B) < box_overlap_share          # the others keep the box beside the path (box_clearance)
            lat_on = side * (hy - rng.uniform(box_overlap[0], box_overlap[1], B))
            gap = 0.3 if box_clearance is None else box_clearance
            lat_off = np.where(lat >= 0.0, 1.0, -1.0) * (np.hypot(hx, hy) + gap)
            th_b = ref[np.arange(B), kb, 2]
            bx = np.where(on, ref[np.arange(B), kb, 0], ref[np.arange(B), kb, 0] - lat_off * np.sin(th_b))
            by = np.where(on, yb + lat_on, yb + lat_off * np.cos(th_b))
        elif box_clearance is not None:   # beside the path: centre beyond the box's circumscribed radius + gap
            side = np.where(lat >= 0.0, 1.0, -1.0)
            lat = side * (np.hypot(hx, hy) + box_clearance)
            bx = ref[np.arange(B), kb, 0] - lat * np.sin(ref[np.arange(B), kb, 2])
            by = ref[np.arange(B), kb, 1] + lat * np.cos(ref[np.arange(B), kb, 2])
        p[:, off["os"] + 12 * o: off["os"] + 12 * (o + 1)] = rect_halfspaces(bx - hx, bx + hx, by - hy, by + hy)
        o += 1

    # ---- dynamic discs crossing the neighbourhood of the path within the horizon
    n_blk = np.zeros(B, dtype=int) if n_block is None else rng.integers(n_block[0], n_block[1] + 1, B)
    for i in range(n_dyn):
        kc = rng.integers(4, N, B)
        lat = rng.uniform(-3.0, 3.0, B)
        if dyn_clearance is not None:   # passing family: lateral offset beyond radius + gap, either side
            lat = np.sign(lat) * (DYN_OBS_SIZE + dyn_clearance + np.abs(lat) / 3.0 * 1.5)
        if n_block is not None:         # avoidance family: the first n_blk discs cover the path, on the side of the nearer wall
            blk_i = i < n_blk
            kcb = rng.integers(min(block_first_step, N - 1), N, B)
            ovl = rng.uniform(block_overlap[0], block_overlap[1], B)
            away = np.where((ref[np.arange(B), kcb, 1] - 5.0) * np.cos(ref[np.arange(B), kcb, 2]) >= 0.0, 1.0, -1.0)
            kc = np.where(blk_i, kcb, kc)
            lat = np.where(blk_i, away * (DYN_OBS_SIZE - ovl), lat)
        jx, jy = rng.uniform(-0.5, 0.5, B), rng.uniform(-0.5, 0.5, B)
        if dyn_clearance is not None:
            jx, jy = 0.0 * jx, 0.0 * jy
        cxp = ref[np.arange(B), kc, 0] - lat * np.sin(ref[np.arange(B), kc, 2]) + jx
        cyp = ref[np.arange(B), kc, 1] + lat * np.cos(ref[np.arange(B), kc, 2]) + jy
        spd = rng.uniform(0.05, 0.3, B)
        dirn = rng.uniform(-np.pi, np.pi, B)
        if dyn_clearance is not None:   # along the path (either way), so the gap holds over the horizon
            dirn = ref[np.arange(B), kc, 2] + np.where(dirn > 0.0, 0.0, np.pi)
        dx, dy = spd * np.cos(dirn), spd * np.sin(dirn)
        ox = cxp[:, None] + dx[:, None] * (ks - kc[:, None])
        oy = cyp[:, None] + dy[:, None] * (ks - kc[:, None])
        # keep the first predicted position clear of the robot
        ex, ey = ox[:, 0] - x, oy[:, 0] - y
        d0 = np.hypot(ex, ey)
        need = DYN_OBS_SIZE + 0.7
        push = np.where(d0 < need, (need - d0) / np.maximum(d0, 1e-9), 0.0)
        ox += (ex * push)[:, None]; oy += (ey * push)[:, None]
        blk = np.zeros((B, N, 6))
        blk[:, :, 0] = ox; blk[:, :, 1] = oy
        blk[:, :, 2] = DYN_OBS_SIZE; blk[:, :, 3] = DYN_OBS_SIZE; blk[:, :, 4] = 0.0; blk[:, :, 5] = 1.0
        p[:, off["od"] + i * 6 * N: off["od"] + (i + 1) * 6 * N] = blk.reshape(B, 6 * N)

    p[:, off["qstc"]:off["qstc"] + N] = 1e3
    if np.ndim(dyn_weight) == 0:
        p[:, off["qdyn"]:off["qdyn"] + N] = float(dyn_weight)
    else:
        p[:, off["qdyn"]:off["qdyn"] + N] = np.exp(rng.uniform(np.log(dyn_weight[0]), np.log(dyn_weight[1]), B))[:, None]
    return dict(p=p, start=np.stack([x, y, th], axis=1), ref=ref)


# Named scene families (tests, bench.py legs, reports).  Every one has the same horizon, table sizes and active-row counts as
# the benchmark family; what changes is where the obstacles stand relative to the reference path.
FAMILIES: Dict[str, dict] = {
    # discs anywhere within 3 m of the path, box on the path: mostly no collision-free plan, solves stop at the caps
    "benchmark": dict(),
    # discs and box beside the path (>= 0.1 / 0.3 m): a collision-free plan exists along the reference
    "passing": dict(dyn_clearance=0.1, box_clearance=0.3),
    # a tick in the middle of a closed-loop run: aligned with a straight reference, at the reference speed
    "on_track": dict(dyn_clearance=0.1, box_clearance=0.3, on_track=True, v_init_range=(1.0, 1.2)),
    # 1-3 discs COVER the path by 0.1-0.6 m (hard radius 1.6 m) with the detour towards the free side of the corridor, the box
    # covers it by 2-30 cm in 30 % of the problems; the robot moves at 0.8-1.2 m/s (a run in progress: with the previous
    # speed far from the reference speed it is the acceleration constraints that stop the ALM loop, not the obstacles).
    # ~45 % of cold-start solves converge at N_hor = 20, > 85 % of those with a positive soft-margin term at the optimum.
    "avoidance": dict(dyn_clearance=0.1, box_clearance=0.3, n_block=(1, 3), block_overlap=(0.1, 0.6),
                      box_overlap=(0.02, 0.3), box_overlap_share=0.3, v_init_range=(0.8, 1.2)),
    # one disc covers the path by 5-50 mm and the tracker passes soft weights of 10 instead of 1e3 (set_obstacle_weights):
    # the plan then rests ON the hard ellipse.  The penalty method needs c >= mu / delta for a multiplier mu, so only small
    # multipliers converge within 10 outer iterations: ~1-2 % of the solves end Converged with F2 > 0 (active hard constraint).
    "grazing": dict(dyn_clearance=0.1, box_clearance=0.3, n_block=(1, 1), block_overlap=(0.005, 0.05), dyn_weight=10.0,
                    v_init_range=(0.8, 1.2)),
}


def make_family(cfg: MpcConfig, B: int, family: str, n_dyn: int = 8, seed: int = 1234, **overrides) -> Dict[str, np.ndarray]:
    """``make_batch`` with the keyword set of a named family (``FAMILIES``)."""
    kw = dict(FAMILIES[family])
    kw.update(overrides)
    return make_batch(cfg, B, n_dyn=n_dyn, seed=seed, **kw)


def shifted_warm_start(u_prev: np.ndarray) -> np.ndarray:
    """Receding-horizon warm start: previous solution shifted by one step, last input repeated."""
    B, n = u_prev.shape
    u = u_prev.reshape(B, n // 2, 2)
    return np.concatenate([u[:, 1:], u[:, -1:]], axis=1).reshape(B, n)
