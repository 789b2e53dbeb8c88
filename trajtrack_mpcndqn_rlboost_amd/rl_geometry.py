"""Polygon preparation for the batched DRL environment (host side, once per map).

The reference pads every obstacle by the robot radius with shapely / GEOS (``Polygon.buffer(0.5, join_style=round,
resolution=4)``, ``src/pkg_dqn/environment/obstacle.py:148-153``) and shrinks the map boundary by the same amount
(``buffer(-0.5, ...)``, ``obstacle.py:229-235``).  shapely is not a dependency of this package, so the offset curve is
built here directly, following GEOS' fillet rule (``OffsetSegmentGenerator::addDirectedFillet``, restated from its
published behaviour):

* a corner that turns away from the offset side gets a circular fillet of ``n = int(angle / (pi / (2 * quad_segs)) +
  0.5)`` equal chords between the end points of the two offset edges (no fillet point when ``n < 1``);
* a corner that turns towards the offset side collapses to the intersection of the two offset edges.

This *local* construction equals the GEOS result whenever the offset ring does not intersect itself (no edge vanishes,
no two distant parts of the outline merge).  That holds for every obstacle / boundary of the reference's maps;
:func:`buffer_polygon` verifies it and raises otherwise instead of returning a wrong outline.
"""
from __future__ import annotations

import math
from typing import Sequence

import numpy as np


def signed_area(ring: np.ndarray) -> float:
    x, y = ring[:, 0], ring[:, 1]
    return 0.5 * float(np.sum(x * np.roll(y, -1) - np.roll(x, -1) * y))


def orient(ring: Sequence[Sequence[float]], ccw: bool = True) -> np.ndarray:
    """Open ring (no repeated last vertex) with the requested orientation; the reference keeps obstacles clockwise and
    the boundary counter-clockwise (``obstacle.py:12-36``) -- only the orientation, not the start vertex, matters."""
    r = np.asarray(ring, dtype=np.float64).reshape(-1, 2)
    if len(r) > 1 and np.allclose(r[0], r[-1]):
        r = r[:-1]
    if (signed_area(r) > 0) != ccw:
        r = r[::-1].copy()
    return r


def _segments_cross(p0, p1, q0, q1) -> bool:
    def side(a, b, c):
        return (b[0] - a[0]) * (c[1] - a[1]) - (b[1] - a[1]) * (c[0] - a[0])
    d1, d2 = side(q0, q1, p0), side(q0, q1, p1)
    d3, d4 = side(p0, p1, q0), side(p0, p1, q1)
    return (d1 * d2 < 0) and (d3 * d4 < 0)


def ring_is_simple(ring: np.ndarray) -> bool:
    n = len(ring)
    for i in range(n):
        for j in range(i + 2, n):
            if i == 0 and j == n - 1:
                continue
            if _segments_cross(ring[i], ring[(i + 1) % n], ring[j], ring[(j + 1) % n]):
                return False
    return True


def buffer_polygon(ring: Sequence[Sequence[float]], distance: float, quad_segs: int = 4, check: bool = True) -> np.ndarray:
    """Offset a simple polygon by ``distance`` (> 0 grows, < 0 shrinks) with round joins; returns the open CCW ring.

    ``quad_segs`` is shapely's ``resolution`` (segments per quarter circle)."""
    poly = orient(ring, ccw=True)
    n = len(poly)
    if n < 3:
        raise ValueError("a polygon needs at least three vertices")
    if distance == 0.0:
        return poly
    r = abs(distance)
    sgn = 1.0 if distance > 0 else -1.0
    quantum = math.pi / 2.0 / quad_segs
    out = []
    for i in range(n):
        v = poly[i]
        d0 = v - poly[i - 1]
        d1 = poly[(i + 1) % n] - v
        d0 = d0 / np.hypot(*d0)
        d1 = d1 / np.hypot(*d1)
        # outward normal of a CCW ring's edge is its right normal; shrinking uses the inward one
        n0 = sgn * np.array([d0[1], -d0[0]])
        n1 = sgn * np.array([d1[1], -d1[0]])
        turn = d0[0] * d1[1] - d0[1] * d1[0]
        if abs(turn) < 1e-14:          # collinear: one offset point
            out.append(v + r * n0)
        elif turn * sgn > 0:           # the corner opens on the offset side: fillet around v
            a0 = math.atan2(n0[1], n0[0])
            total = math.acos(max(-1.0, min(1.0, float(n0 @ n1))))
            nseg = int(total / quantum + 0.5)
            out.append(v + r * n0)
            if nseg >= 1:
                inc = total / nseg * (1.0 if turn > 0 else -1.0)
                for k in range(1, nseg):
                    out.append(v + r * np.array([math.cos(a0 + k * inc), math.sin(a0 + k * inc)]))
            out.append(v + r * n1)
        else:                          # the offset edges meet: their intersection
            out.append(v + r * (n0 + n1) / (1.0 + float(n0 @ n1)))
    res = np.asarray(out)
    # drop (near-)duplicate consecutive points, as GEOS' minimum vertex distance does
    keep = [0]
    for i in range(1, len(res)):
        if np.hypot(*(res[i] - res[keep[-1]])) > 1e-6 * r:
            keep.append(i)
    if len(keep) > 1 and np.hypot(*(res[keep[-1]] - res[keep[0]])) <= 1e-6 * r:
        keep.pop()
    res = res[keep]
    if check:
        if signed_area(res) <= 0 or not ring_is_simple(res):
            raise ValueError("offset ring intersects itself: the local offset construction does not apply to this polygon")
        if distance < 0 and not all(point_in_ring(p, poly) for p in res):
            raise ValueError("shrunk ring leaves the polygon: the local offset construction does not apply")
    return res


def point_in_ring(pt, ring: np.ndarray) -> bool:
    """Even-odd rule (points exactly on the outline are a measure-zero case the reference does not rely on)."""
    x, y = float(pt[0]), float(pt[1])
    inside = False
    n = len(ring)
    for i in range(n):
        x0, y0 = ring[i]
        x1, y1 = ring[(i + 1) % n]
        if (y0 > y) != (y1 > y) and x < (x1 - x0) * (y - y0) / (y1 - y0) + x0:
            inside = not inside
    return inside


def ellipse_nodes(rx: float, ry: float, corners: int = 12) -> np.ndarray:
    """Body-frame polygon of a dynamic obstacle: ``(rx cos a, -ry sin a)``, ``a = 2 pi i / corners``
    (``obstacle.py:193-199``)."""
    a = 2.0 * math.pi * np.arange(corners) / corners
    return np.stack([rx * np.cos(a), -ry * np.sin(a)], axis=1)
