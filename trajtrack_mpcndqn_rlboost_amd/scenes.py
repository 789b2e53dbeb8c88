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
               on_track: bool = False) -> Dict[str, np.ndarray]:
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
    looks like) instead of up to 0.45 rad off a reference with a corner."""
    N = int(cfg.N_hor)
    off = cfg.offsets()
    assert n_dyn <= cfg.Ndynobs and n_other <= cfg.Nother
    rng = np.random.default_rng(seed)
    p = np.zeros((B, cfg.num_params))
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
        if box_clearance is not None:   # beside the path: centre beyond the box's circumscribed radius + gap
            side = np.where(lat >= 0.0, 1.0, -1.0)
            lat = side * (np.hypot(hx, hy) + box_clearance)
            bx = ref[np.arange(B), kb, 0] - lat * np.sin(ref[np.arange(B), kb, 2])
            by = ref[np.arange(B), kb, 1] + lat * np.cos(ref[np.arange(B), kb, 2])
        p[:, off["os"] + 12 * o: off["os"] + 12 * (o + 1)] = rect_halfspaces(bx - hx, bx + hx, by - hy, by + hy)
        o += 1

    # ---- dynamic discs crossing the neighbourhood of the path within the horizon
    for i in range(n_dyn):
        kc = rng.integers(4, N, B)
        lat = rng.uniform(-3.0, 3.0, B)
        if dyn_clearance is not None:   # passing family: lateral offset beyond radius + gap, either side
            lat = np.sign(lat) * (DYN_OBS_SIZE + dyn_clearance + np.abs(lat) / 3.0 * 1.5)
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
    p[:, off["qdyn"]:off["qdyn"] + N] = 1e3
    return dict(p=p, start=np.stack([x, y, th], axis=1), ref=ref)


def shifted_warm_start(u_prev: np.ndarray) -> np.ndarray:
    """Receding-horizon warm start: previous solution shifted by one step, last input repeated."""
    B, n = u_prev.shape
    u = u_prev.reshape(B, n // 2, 2)
    return np.concatenate([u[:, 1:], u[:, -1:]], axis=1).reshape(B, n)
