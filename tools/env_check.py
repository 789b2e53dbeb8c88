"""Quick GPU check + throughput of the environment kernel: python tools/env_check.py [B] [steps]"""
import importlib, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rl_env = importlib.import_module("trajtrack_mpcndqn_rlboost_amd.rl_env")
fx = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "env_rays_traces.npz"))
specs = json.loads(bytes(fx["specs_json"]).decode())
maps = [rl_env.make_map(sp["boundary"], sp["static"], sp["dynamic"], sp["start"], sp["goal"], sp["path"]) for sp in specs.values()]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
env = rl_env.BatchedRaysEnv([maps[i % 2] for i in range(B)])
env.reset()
acts = torch.randint(0, 9, (steps, B), device=env.device, dtype=torch.int32)
torch.cuda.synchronize()
t0 = time.perf_counter()
for t in range(steps):
    env._launch(acts[t])
torch.cuda.synchronize()
dt = time.perf_counter() - t0
rec_bytes = env.records.shape[1] * 8
print(f"B={B} steps={steps}: {B * steps / dt:.3e} env-steps/s, {dt / steps * 1e3:.3f} ms/step, "
      f"record {rec_bytes} B/env -> {B * rec_bytes * steps / dt / 1e9:.1f} GB/s of map reads")
