"""Unicycle motion model (numpy), the one model the tracker is compiled for.

Reference: ``src/pkg_motion_model/motion_model.py:142-164`` (RK4 of x' = v cos(theta), y' = v sin(theta),
theta' = w with the inputs held over the step).  Because theta' does not depend on the state, the four RK4
stages see only three distinct headings and the step collapses to Simpson's rule on the heading
 -- that closed form is what the GPU kernel integrates; here the four stages are kept
so that the host-side rollouts reproduce the reference's numbers stage by stage.
"""
from __future__ import annotations

import numpy as np


def _increment(state: np.ndarray, action: np.ndarray, ts: float) -> np.ndarray:
    heading = state[..., 2]
    return ts * np.stack([action[..., 0] * np.cos(heading), action[..., 0] * np.sin(heading), action[..., 1]], axis=-1)


def unicycle_model(state: np.ndarray, action: np.ndarray, ts: float, rk4: bool = True) -> np.ndarray:
    """Next state for ``state = (x, y, theta)``, ``action = (v, w)``.  Leading batch dimensions broadcast."""
    state = np.asarray(state, dtype=float)
    action = np.asarray(action, dtype=float)
    k1 = _increment(state, action, ts)
    if not rk4:
        return state + k1
    k2 = _increment(state + 0.5 * k1, action, ts)
    k3 = _increment(state + 0.5 * k2, action, ts)
    k4 = _increment(state + k3, action, ts)
    return state + (1 / 6) * (k1 + 2 * k2 + 2 * k3 + k4)
