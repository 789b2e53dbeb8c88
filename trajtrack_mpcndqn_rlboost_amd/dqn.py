"""DQN action-value head and the RL reference it proposes to the MPC (hybrid mode of the reference).

* ``QNetwork``: the ray-observation model of the reference -- SB3 ``DQN('MultiInputPolicy')`` with
  ``net_arch=[16, 16]``: 46 -> 16 -> 16 -> 9, ReLU (``src/test_block_rl.py:40-53,77-86``; input = ``external``(32)
  then ``internal``(14), dict keys sorted).  Plain ``torch.nn`` on PyTorch-ROCm: 1 177 parameters, no custom kernel.
* ``rl_reference``: what ``src/main.py:193-202`` does with the chosen action -- copy the agent, apply the action for
  one step (``src/pkg_dqn/environment/agent.py:102-145``), then 19 steps at reference speed 1.0 with the angular
  velocity decaying by 0.95 per step (``agent.py:86-100``) -- vectorised over the batch.
* ``merge_reference``: ``InterfaceMpc.get_local_ref_traj(rl_ref)`` appends the heading column of the original
  reference (``src/interface_mpc.py:77-79``).
"""
from __future__ import annotations

import io
import zipfile
from typing import Dict

import numpy as np
import torch
from torch import nn

# agent.py:7-16
SPEED_MIN, SPEED_MAX = -0.5, 1.5
ANGULAR_VELOCITY_MIN, ANGULAR_VELOCITY_MAX = -0.5, 0.5
ACCELERATION_MIN, ACCELERATION_MAX = -1.0, 1.0
ANGULAR_ACCELERATION_MIN, ANGULAR_ACCELERATION_MAX = -3.0, 3.0
OBS_DIM, N_ACTIONS = 46, 9


class QNetwork(nn.Module):
    def __init__(self, obs_dim: int = OBS_DIM, hidden=(16, 16), n_actions: int = N_ACTIONS):
        super().__init__()
        layers, d = [], obs_dim
        for h in hidden:
            layers += [nn.Linear(d, h), nn.ReLU()]
            d = h
        layers.append(nn.Linear(d, n_actions))
        self.q_net = nn.Sequential(*layers)

    def forward(self, obs: torch.Tensor) -> torch.Tensor:
        return self.q_net(obs)

    @torch.no_grad()
    def greedy_actions(self, obs: torch.Tensor) -> torch.Tensor:
        """``model.predict(obs, deterministic=True)`` for a batch (src/main.py:184)."""
        return self.forward(obs).argmax(dim=1)

    def load_arrays(self, arrays: Dict[str, np.ndarray]) -> "QNetwork":
        """arrays: {'w0_weight','w0_bias','w2_weight','w2_bias','w4_weight','w4_bias'} (SB3 layer indices)."""
        with torch.no_grad():
            for idx in (0, 2, 4):
                self.q_net[idx].weight.copy_(torch.as_tensor(arrays[f"w{idx}_weight"]))
                self.q_net[idx].bias.copy_(torch.as_tensor(arrays[f"w{idx}_bias"]))
        return self

    @classmethod
    def from_sb3_zip(cls, path: str) -> "QNetwork":
        """Read ``policy.pth`` out of an SB3 ``model.save`` archive (e.g. the reference's Model/ray/best_model.zip)."""
        with zipfile.ZipFile(path) as z:
            sd = torch.load(io.BytesIO(z.read("policy.pth")), weights_only=True)
        return cls().load_arrays({f"w{i}_{k}": sd[f"q_net.q_net.{i}.{k}"].numpy() for i in (0, 2, 4)
                                  for k in ("weight", "bias")})


def observation_vector(obs: Dict[str, np.ndarray]) -> np.ndarray:
    """SB3's CombinedExtractor concatenates the dict entries in sorted key order: 'external' then 'internal'."""
    return np.concatenate([np.asarray(obs[k], dtype=np.float32).reshape(len(obs[k]) if np.ndim(obs[k]) > 1 else 1, -1)
                           for k in sorted(obs)], axis=1)


def rl_reference(states: np.ndarray, action_index: np.ndarray, ts: float, steps: int = 20, ref_speed: float = 1.0):
    """states [B, 5] = (x, y, theta, v, w); action_index [B] in 0..8.  Returns (rl_ref [B, steps, 2], final states)."""
    s = np.array(states, dtype=float, copy=True)
    a = np.asarray(action_index)
    x, y, th, v, w = (s[:, i].copy() for i in range(5))
    # step 0: the chosen acceleration pair (agent.py:124-145)
    v = v + ts * np.where(a // 3 == 0, ACCELERATION_MAX, 0.0) + ts * np.where(a // 3 == 2, ACCELERATION_MIN, 0.0)
    w = w + ts * np.where(a % 3 == 0, ANGULAR_ACCELERATION_MAX, 0.0) + ts * np.where(a % 3 == 2, ANGULAR_ACCELERATION_MIN, 0.0)
    v = np.clip(v, SPEED_MIN, SPEED_MAX)
    w = np.clip(w, ANGULAR_VELOCITY_MIN, ANGULAR_VELOCITY_MAX)
    th = th + ts * w
    x = x + ts * v * np.cos(th)
    y = y + ts * v * np.sin(th)
    out = np.empty((len(s), steps, 2))
    out[:, 0, 0], out[:, 0, 1] = x, y
    speed = ref_speed if ref_speed > 0.0 else SPEED_MAX
    for j in range(1, steps):                      # agent.py:86-100
        w = w * 0.95
        th = th + ts * w
        x = x + ts * speed * np.cos(th)
        y = y + ts * speed * np.sin(th)
        out[:, j, 0], out[:, j, 1] = x, y
    return out, np.stack([x, y, th, v, w], axis=1)


def merge_reference(rl_ref_xy: np.ndarray, original_ref: np.ndarray) -> np.ndarray:
    """[.., N, 2] RL reference + heading column of the original local reference -> [.., N, 3]."""
    return np.concatenate([rl_ref_xy, original_ref[..., 2:3]], axis=-1)
