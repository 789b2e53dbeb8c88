"""Dynamic-obstacle prediction feeders: what the callers of the tracker put into the dynamic-obstacle block.

* ``constant_velocity_prediction``: the reference's ``est_dyn_obs_positions`` (``src/main.py:77-85``) -- position
  extrapolated with the last displacement, fixed disc size ``DYN_OBS_SIZE = 0.8 + 0.8`` (``src/main.py:31``),
  angle 0, alpha 1 -- vectorised over obstacles and for any horizon (the reference hard-codes 20 steps).
* ``scanner_prediction``: the format of the scripted multimodal scanners
  (``src/obstacle_simulator/_obstacle_simulator.py:48-76``): per mode and step ``(alpha, x, y, sx, sy, angle)`` is
  re-ordered to ``(x, y, sx*factor + r, sy*factor + r, angle, alpha)``.
Both return arrays shaped ``[n_obstacles, N, 6]`` that ``InterfaceMpc.update_dynamic_constraints`` /
``BatchedTracker.update_dynamic_constraints`` accept directly.
"""
from __future__ import annotations

import numpy as np

DYN_OBS_SIZE = 0.8 + 0.8


def constant_velocity_prediction(last_pos, current_pos, steps: int = 20, size: float = DYN_OBS_SIZE) -> np.ndarray:
    """last_pos, current_pos: [..., 2].  Returns [..., steps, 6] = (x, y, size, size, 0, 1) at current + (k+1)*delta."""
    last_pos = np.asarray(last_pos, dtype=float)
    current_pos = np.asarray(current_pos, dtype=float)
    delta = current_pos - last_pos
    k = np.arange(1, steps + 1, dtype=float)
    out = np.zeros(current_pos.shape[:-1] + (steps, 6))
    out[..., 0] = current_pos[..., None, 0] + delta[..., None, 0] * k
    out[..., 1] = current_pos[..., None, 1] + delta[..., None, 1] * k
    out[..., 2] = size
    out[..., 3] = size
    out[..., 5] = 1.0
    return out


def scanner_prediction(pred, inflation_radius: float, factor: float = 1.0) -> np.ndarray:
    """pred: [n_modes, T, 6] rows (alpha, x, y, sx, sy, angle) -> [n_modes, T, 6] rows (x, y, rx, ry, angle, alpha)."""
    pred = np.asarray(pred, dtype=float)
    out = np.empty_like(pred)
    out[..., 0] = pred[..., 1]
    out[..., 1] = pred[..., 2]
    out[..., 2] = pred[..., 3] * factor + inflation_radius
    out[..., 3] = pred[..., 4] * factor + inflation_radius
    out[..., 4] = pred[..., 5]
    out[..., 5] = pred[..., 0]
    return out
