#!/usr/bin/env python3
"""Golden vectors for the hybrid DQN -> MPC reference generator.

Run in the build container only (needs /root/reference):   python tests/golden/make_dqn_fixtures.py

* rollouts: the reference's own ``MobileRobot`` (src/pkg_dqn/environment/agent.py:19-145) is executed the way
  src/main.py:193-202 does -- step 0 with the DQN action index, steps 1..19 with ``step_with_ref_speed(ts, 1.0)``
  -- on seeded random states (shapely, which agent.py imports only for a ``Point`` attribute, is stubbed).
* Q-values: the reference's trained ray model (Model/ray/best_model.zip, member policy.pth, SB3 MlpPolicy
  46->16->16->9 with ReLU: src/test_block_rl.py:40-53,77-86) evaluated with plain torch on seeded observations.
  The 1 177 weights are stored next to the expected outputs because SB3 / the zip are not available on the GPU box.
Only data (inputs, weights, expected outputs) is written: dqn_ray.npz.
"""
import copy
import io
import os
import sys
import types
import zipfile

import numpy as np
import torch

sys.dont_write_bytecode = True
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

shapely = types.ModuleType("shapely")
geom = types.ModuleType("shapely.geometry")
geom.Point = lambda *a, **k: None
shapely.geometry = geom
sys.modules["shapely"] = shapely
sys.modules["shapely.geometry"] = geom

import importlib.util  # noqa: E402
spec = importlib.util.spec_from_file_location("ref_agent", os.path.join(REF, "src/pkg_dqn/environment/agent.py"))
ref_agent = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref_agent)

if __name__ == "__main__":
    rng = np.random.default_rng(2026)
    B, STEPS, TS = 96, 20, 0.2
    states = np.stack([rng.uniform(0, 15, B), rng.uniform(0, 10, B), rng.uniform(-np.pi, np.pi, B),
                       rng.uniform(-0.5, 1.5, B), rng.uniform(-0.5, 0.5, B)], axis=1)
    actions = rng.integers(0, 9, B)
    ref = np.zeros((B, STEPS, 2))
    final = np.zeros((B, 5))
    for i in range(B):
        robot = ref_agent.MobileRobot(states[i].copy())
        sim = copy.deepcopy(robot)
        for j in range(STEPS):                      # src/main.py:195-202
            if j == 0:
                sim.step(int(actions[i]), TS)
            else:
                sim.step_with_ref_speed(TS, 1.0)
            ref[i, j] = list(sim.position)
        final[i] = sim.state
    z = zipfile.ZipFile(os.path.join(REF, "Model/ray/best_model.zip"))
    sd = torch.load(io.BytesIO(z.read("policy.pth")), weights_only=True)
    W = {k.replace("q_net.q_net.", "w"): v.numpy() for k, v in sd.items() if k.startswith("q_net.q_net.")}
    obs = rng.uniform(-1.0, 1.0, (256, 46)).astype(np.float32)      # [external(32); internal(14)], keys sorted
    with torch.no_grad():
        x = torch.from_numpy(obs)
        x = torch.relu(x @ sd["q_net.q_net.0.weight"].T + sd["q_net.q_net.0.bias"])
        x = torch.relu(x @ sd["q_net.q_net.2.weight"].T + sd["q_net.q_net.2.bias"])
        q = x @ sd["q_net.q_net.4.weight"].T + sd["q_net.q_net.4.bias"]
    np.savez_compressed(os.path.join(HERE, "dqn_ray.npz"), states=states, actions=actions, ts=TS, rl_ref=ref,
                        final_state=final, obs=obs, q=q.numpy(), greedy=q.argmax(dim=1).numpy(),
                        **{k.replace(".", "_"): v for k, v in W.items()})
    print("written", {k: v.shape for k, v in W.items()}, "greedy histogram", np.bincount(q.argmax(dim=1).numpy(), minlength=9))
