"""GPU parity tests for the batched DRL environment kernel: ``mpcgpu_env_step_dev``
against (i) traces of the reference's own environment code (tests/golden/env_rays_traces.npz) and (ii) the CPU
oracle on random robot states / clock values.  Tolerances: robot state 1e-12 (same float64 operations, FMA
contraction allowed), rewards 1e-9, observations 2e-6 (the reference rounds them through float32), flags exact."""
import importlib
import json
import os

import numpy as np
import pytest

from oracle import rl_env_numpy as orc

pytestmark = pytest.mark.gpu
rl_env = importlib.import_module("trajtrack_mpcndqn_rlboost_amd.rl_env")
GOLD = os.path.join(os.path.dirname(__file__), "golden", "env_rays_traces.npz")
KEYS = ["scene1_r0", "scene1_r1", "lhall_r0", "lhall_r1"]


def load():
    fx = np.load(GOLD)
    specs = json.loads(bytes(fx["specs_json"]).decode())
    maps = {name: rl_env.make_map(sp["boundary"], sp["static"], sp["dynamic"], sp["start"], sp["goal"], sp["path"])
            for name, sp in specs.items()}
    return fx, maps


@pytest.mark.parametrize("ts_run", [0, 1])
def test_kernel_reproduces_reference_traces_in_one_batch(ts_run):
    """Both scenes step side by side in ONE launch per step (different maps, edge counts, obstacle counts)."""
    import torch
    fx, maps = load()
    keys = [k for k in KEYS if k.endswith(f"r{ts_run}")]
    env = rl_env.BatchedRaysEnv([maps[k.split("_")[0]] for k in keys], time_step=float(fx[keys[0] + "_ts"]))
    obs = env.reset()
    steps = min(len(fx[k + "_actions"]) for k in keys)
    for b, k in enumerate(keys):
        assert np.abs(obs["internal"][b].cpu().numpy() - fx[k + "_internal"][0]).max() <= 1e-6
        assert np.abs(obs["external"][b].cpu().numpy() - fx[k + "_external"][0]).max() <= 2e-6
    for t in range(steps):
        acts = torch.tensor([int(fx[k + "_actions"][t]) for k in keys])
        obs, rew, term, trunc, info = env.step(acts)
        st = env.agent_state.cpu().numpy()
        fl = env.flags.cpu().numpy()
        for b, k in enumerate(keys):
            assert np.abs(st[b] - fx[k + "_state"][t + 1]).max() <= 1e-12, (k, t)
            assert np.array_equal(fl[b], fx[k + "_flags"][t]), (k, t)
            assert bool(term[b]) == bool(fx[k + "_done"][t])
            assert abs(float(rew[b]) - fx[k + "_reward"][t]) <= 1e-9, (k, t)
            assert abs(float(env.path_progress[b]) - fx[k + "_progress"][t + 1]) <= 1e-12
            assert np.abs(obs["internal"][b].cpu().numpy() - fx[k + "_internal"][t + 1]).max() <= 1e-6, (k, t)
            assert np.abs(obs["external"][b].cpu().numpy() - fx[k + "_external"][t + 1]).max() <= 2e-6, (k, t)
            assert bool(info["success"][b]) == bool(fx[k + "_flags"][t][2])
    # observe-only path (main.py:181-189) after the trace, memory included
    if steps == min(len(fx[k + "_actions"]) for k in KEYS if k.endswith(f"r{ts_run}")) and all(len(fx[k + "_actions"]) == steps for k in keys):
        for j in range(2):
            env.set_agent_state(np.stack([fx[k + "_teleport"][j] for k in keys]))
            o = env.observe()
            for b, k in enumerate(keys):
                assert np.abs(o["internal"][b].cpu().numpy() - fx[k + "_tele_internal"][j]).max() <= 1e-6
                assert np.abs(o["external"][b].cpu().numpy() - fx[k + "_tele_external"][j]).max() <= 2e-6


@pytest.mark.parametrize("name", ["scene1", "lhall"])
def test_kernel_matches_oracle_on_random_states_and_clock_values(name):
    """512 random robot poses (inside / outside obstacles and walls) at random obstacle clock values: the per-edge
    wedge clipping of the kernel against the oracle's polygon clipping."""
    import torch
    _, maps = load()
    m = maps[name]
    rng = np.random.default_rng(5)
    B = 512
    ring = np.asarray(m["boundary_padded"])
    lo, hi = ring.min(axis=0) - 0.4, ring.max(axis=0) + 0.4
    states = np.stack([rng.uniform(lo[0], hi[0], B), rng.uniform(lo[1], hi[1], B), rng.uniform(-4, 4, B),
                       rng.uniform(-0.5, 1.5, B), rng.uniform(-0.5, 0.5, B)], axis=1)
    clocks = rng.uniform(0, 60, B)
    env = rl_env.BatchedRaysEnv([m] * B)
    env.reset()
    env.state[:, 8:24] = 0.0
    env.state[:, 7] = 0.0
    env.set_agent_state(states)
    env.state[:, 5] = torch.from_numpy(clocks).to(env.device)
    obs = env.observe()
    oi, oe = obs["internal"].cpu().numpy(), obs["external"].cpu().numpy()
    fl = env.flags.cpu().numpy()
    prog = env.path_progress.cpu().numpy()
    n_inside = 0
    for b in range(0, B, 2):
        o = orc.OracleRaysEnv(m)
        o.old_obs[:] = 0.0
        o.state[:] = states[b]
        o.time = clocks[b]
        o.collided_obstacle = o.collided_boundary = o.collided = o.reached_goal = False
        ob, _, done, _ = o.step(None)
        assert np.abs(oi[b] - ob["internal"]).max() <= 1e-6, b
        assert np.abs(oe[b] - ob["external"]).max() <= 2e-6, (b, oe[b], ob["external"])
        assert np.array_equal(fl[b], [o.collided_obstacle, o.collided_boundary, o.reached_goal]), b
        assert abs(prog[b] - o.progress) <= 1e-12
        n_inside += int(o.collided)
    assert 10 < n_inside < B // 2 - 10   # both branches are exercised


def test_auto_reset_truncation_and_masked_reset():
    import torch
    _, maps = load()
    env = rl_env.BatchedRaysEnv([maps["scene1"]] * 8, max_episode_steps=5)
    first = env.reset()
    start = env.agent_state.clone()
    acts = torch.full((8,), 1)
    for t in range(5):
        obs, rew, term, trunc, info = env.step(acts, auto_reset=True)
    # five steps without collision: truncated by the step limit, then back at the start state
    assert not bool(term.any()) and bool(trunc.all())
    assert torch.equal(env.agent_state, start)
    assert "terminal_observation" in info
    assert torch.allclose(obs["internal"], first["internal"])
    # the observation memory survives the reset, as the reference's component keeps old_obs (no reset() override)
    assert torch.equal(obs["external"][:, 16:], info["terminal_observation"]["external"][:, :16])
    # masked reset leaves the other rows untouched
    env.step(acts)
    before = env.state.clone()
    mask = torch.tensor([True, False] * 4)
    env.reset(mask)
    assert torch.equal(env.state[~mask.to(env.device)], before[~mask.to(env.device)])
    assert torch.equal(env.agent_state[mask.to(env.device)], start[mask.to(env.device)])


def test_invalid_calls_fail_loudly():
    import torch
    _, maps = load()
    env = rl_env.BatchedRaysEnv([maps["lhall"]] * 3)
    with pytest.raises(ValueError):
        env.step(torch.zeros(4, dtype=torch.int64))
    env.params.num_segments = 16
    with pytest.raises(rl_env.MpcGpuError, match="num_segments"):
        env.observe()


def _random_map(rng):
    """A random hall with random convex obstacles on general key-frame animations (1..4 key frames, linear or cosine
    easing, time offsets) -- exercises what the two fixture scenes do not: n_kf_max = 4, linear interpolation, offsets,
    different obstacle / edge / path-node counts per environment in one batch."""
    import math
    rg = importlib.import_module("trajtrack_mpcndqn_rlboost_amd.rl_geometry")
    W, H = rng.uniform(12, 30), rng.uniform(10, 25)
    boundary = [(0, 0), (W, 0), (W, H), (0, H)]
    if rng.random() < 0.5:                      # notch: a reflex corner in the boundary
        boundary = [(0, 0), (W, 0), (W, H * 0.6), (W * 0.7, H * 0.6), (W * 0.7, H), (0, H)]
    obstacles = []
    for _ in range(rng.integers(0, 7)):
        n = rng.integers(3, 7)
        ang = np.sort(rng.uniform(0, 2 * math.pi, n))
        if np.min(np.diff(np.concatenate([ang, [ang[0] + 2 * math.pi]]))) < 0.4:
            continue
        rad = rng.uniform(0.5, 2.0)
        nodes = np.stack([rad * np.cos(ang), rad * np.sin(ang)], axis=1)
        if rg.signed_area(nodes) < 0.3:
            continue
        nk = int(rng.integers(1, 5))
        frames = [(rng.uniform(1, W - 1), rng.uniform(1, H - 1), rng.uniform(-3, 3)) for _ in range(nk)]
        steps = [0.0] + [float(rng.uniform(0.5, 6.0)) for _ in range(nk)]
        obstacles.append(dict(padded_nodes=rg.buffer_polygon(nodes, 0.5), time_steps=steps, keyframes=frames,
                              interp="cosine" if rng.random() < 0.5 else "linear", offset=float(rng.uniform(0, 5))))
    npath = int(rng.integers(2, 9))
    path = np.stack([np.sort(rng.uniform(0.5, W - 0.5, npath)), rng.uniform(0.5, H * 0.55, npath)], axis=1)
    return dict(start=np.array([path[0, 0], path[0, 1], 0.0, 0.0, 0.0]), goal=np.asarray(path[-1], dtype=np.float32).astype(float),
                path=path, boundary_padded=rg.buffer_polygon(boundary, -0.5), obstacles=obstacles)


def test_random_maps_with_general_keyframe_animations_match_the_oracle():
    import torch
    rng = np.random.default_rng(99)
    maps = [_random_map(rng) for _ in range(96)]
    assert max(len(o["keyframes"]) for m in maps for o in m["obstacles"]) == 4
    env = rl_env.BatchedRaysEnv(maps, time_step=0.15)
    B = len(maps)
    oracles = [orc.OracleRaysEnv(m, time_step=0.15) for m in maps]
    env.reset()
    for k in range(12):
        acts = rng.integers(0, 9, B)
        if k % 4 == 3:                              # teleport everybody, anywhere in (or slightly outside) the hall
            st = np.zeros((B, 5))
            for b, m in enumerate(maps):
                ring = np.asarray(m["boundary_padded"])
                st[b] = [rng.uniform(ring[:, 0].min() - 0.3, ring[:, 0].max() + 0.3),
                         rng.uniform(ring[:, 1].min() - 0.3, ring[:, 1].max() + 0.3), rng.uniform(-4, 4),
                         rng.uniform(-0.5, 1.5), rng.uniform(-0.5, 0.5)]
                oracles[b].state[:] = st[b]
            env.set_agent_state(st)
        obs, rew, term, _, _ = env.step(torch.from_numpy(acts))
        oi, oe = obs["internal"].cpu().numpy(), obs["external"].cpu().numpy()
        st, fl = env.agent_state.cpu().numpy(), env.flags.cpu().numpy()
        for b, o in enumerate(oracles):
            ob, r, done, _ = o.step(int(acts[b]))
            assert np.abs(st[b] - o.state).max() <= 1e-12, (k, b)
            assert np.array_equal(fl[b], [o.collided_obstacle, o.collided_boundary, o.reached_goal]), (k, b)
            assert np.abs(oi[b] - ob["internal"]).max() <= 1e-6, (k, b)
            assert np.abs(oe[b] - ob["external"]).max() <= 2e-6, (k, b, oe[b], ob["external"])
            assert abs(float(rew[b]) - r) <= 1e-9 and bool(term[b]) == done


def test_degenerate_maps_no_obstacles_and_longest_path():
    """M = 0 (boundary only) and the 64-node path limit, both against the oracle."""
    import torch
    rg = importlib.import_module("trajtrack_mpcndqn_rlboost_amd.rl_geometry")
    xs = np.linspace(1.0, 29.0, 64)
    long_path = np.stack([xs, 5.0 + 2.0 * np.sin(xs / 3.0)], axis=1)
    for path in ([(1.0, 1.0), (9.0, 9.0)], long_path):
        m = dict(start=np.array([path[0][0], path[0][1], 0.3, 0.5, 0.0]), goal=np.asarray(path[-1], dtype=float),
                 path=np.asarray(path, dtype=float), boundary_padded=rg.buffer_polygon([(0, 0), (30, 0), (30, 10), (0, 10)], -0.5),
                 obstacles=[])
        env = rl_env.BatchedRaysEnv([m, m])
        o = orc.OracleRaysEnv(m)
        env.reset()
        rng = np.random.default_rng(0)
        for _ in range(40):
            a = int(rng.integers(0, 9))
            obs, rew, term, _, _ = env.step(torch.tensor([a, a]))
            ob, r, done, _ = o.step(a)
            assert np.abs(env.agent_state[1].cpu().numpy() - o.state).max() <= 1e-12
            assert np.abs(obs["internal"][1].cpu().numpy() - ob["internal"]).max() <= 1e-6
            assert np.abs(obs["external"][1].cpu().numpy() - ob["external"]).max() <= 2e-6
            assert abs(float(rew[1]) - r) <= 1e-9 and bool(term[1]) == done
            assert abs(float(env.path_progress[0]) - o.progress) <= 1e-12
    with pytest.raises(ValueError):
        rl_env.pack_records([dict(m, path=np.zeros((65, 2)))])


def test_in_kernel_auto_reset_matches_oracle_reset_semantics():
    """step(auto_reset=True): the episode bookkeeping done inside the kernel (second observation pass after the reset)
    against the oracle driven the gym way -- step, and on termination / time limit keep the terminal observation,
    reset(), continue.  Random actions make the robots collide every few dozen steps."""
    import torch
    _, maps = load()
    m = maps["lhall"]
    B, LIMIT = 6, 40
    env = rl_env.BatchedRaysEnv([m] * B, max_episode_steps=LIMIT)
    oracles = [orc.OracleRaysEnv(m) for _ in range(B)]
    env.reset()
    rng = np.random.default_rng(8)
    steps = np.zeros(B, dtype=int)
    ended = {"terminated": 0, "truncated": 0}
    for t in range(150):
        acts = rng.integers(0, 9, B)
        acts[0] = 1 if t % 7 else 4                      # robot 0 mostly drives straight: it ends by collision
        acts[1] = 7                                      # robot 1 brakes / reverses slowly: it ends by the time limit
        obs, rew, term, trunc, info = env.step(torch.from_numpy(acts), auto_reset=True)
        for b, o in enumerate(oracles):
            ob, r, done, oinfo = o.step(int(acts[b]))
            steps[b] += 1
            timeout = (not done) and steps[b] >= LIMIT
            assert bool(term[b]) == done and bool(trunc[b]) == timeout, (t, b)
            assert abs(float(rew[b]) - r) <= 1e-9
            assert bool(info["success"][b]) == oinfo["success"]
            tob = {k: info["terminal_observation"][k][b].cpu().numpy() for k in ("internal", "external")}
            assert np.abs(tob["internal"] - ob["internal"]).max() <= 1e-6 and np.abs(tob["external"] - ob["external"]).max() <= 2e-6
            if done or timeout:
                ended["terminated" if done else "truncated"] += 1
                ob = o.reset()
                steps[b] = 0
            assert np.abs(obs["internal"][b].cpu().numpy() - ob["internal"]).max() <= 1e-6, (t, b)
            assert np.abs(obs["external"][b].cpu().numpy() - ob["external"]).max() <= 2e-6, (t, b)
            assert np.abs(env.agent_state[b].cpu().numpy() - o.state).max() <= 1e-12
    assert ended["terminated"] >= 3 and ended["truncated"] >= 2
