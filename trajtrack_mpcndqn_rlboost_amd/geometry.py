"""Half-space (H-) representation of convex polygons: the static-obstacle block of the parameter vector.

Reference: ``src/util/utils_geo.py:33-59`` -- every facet of the convex hull is written as ``a . (x - c) = 1`` with
``c`` the mean of the hull vertices, i.e. the polygon is ``{x : b - a0 x - a1 y > 0}`` with ``b = a . c + 1``
(``src/mpc_traj_tracker/mpc/mpc_generator.py:46-54`` consumes exactly that).  Here the facets come from Qhull's
own facet equations ``n . x + d <= 0`` (same facet order as ``hull.simplices``, which the reference loops over):
with ``s = -(d + n . c)`` the reference's normalisation is ``a = n / s`` and ``b = -d / s``.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np
from scipy.spatial import ConvexHull


def polygon_halfspace_representation(polygon_points: np.ndarray) -> Tuple[List[float], List[float], List[float]]:
    pts = np.asarray(polygon_points, dtype=float)
    hull = ConvexHull(pts)
    centre = pts[hull.vertices].mean(axis=0)
    normals, offsets = hull.equations[:, :2], hull.equations[:, 2]
    scale = -(offsets + normals @ centre)          # distance-like factor of every facet from the centre (> 0)
    keep = np.abs(scale) > 1e-14                    # a facet through the centre has no such representation
    a = normals[keep] / scale[keep, None]
    b = -offsets[keep] / scale[keep]
    return b.tolist(), a[:, 0].tolist(), a[:, 1].tolist()


def static_obstacle_params(obstacle_list: Sequence[Sequence[Sequence[float]]], n_slots: int,
                           n_per_obstacle: int = 12) -> List[float]:
    """Padded static-obstacle block (``src/interface_mpc.py:23,60-63``).  Unlike the reference, an obstacle whose
    hull does not have exactly ``n_per_obstacle / 3`` edges is an error instead of silently changing the length of
    the block."""
    if len(obstacle_list) > n_slots:
        raise ValueError(f"{len(obstacle_list)} static obstacles but only {n_slots} slots")
    block = [0.0] * (n_slots * n_per_obstacle)
    for i, poly in enumerate(obstacle_list):
        b, a0, a1 = polygon_halfspace_representation(np.array(poly, dtype=float))
        if 3 * len(b) != n_per_obstacle:
            raise ValueError(f"static obstacle {i} has {len(b)} hull edges; the solver is built for {n_per_obstacle // 3}")
        block[i * n_per_obstacle:(i + 1) * n_per_obstacle] = b + a0 + a1
    return block
