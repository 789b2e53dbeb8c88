"""Probe (GPU, -DMPC_TRACE variant): how many line-search trials and Lipschitz retries a PANOC step takes.
usage: python tools/probes/ls_distribution.py [N = 20] [B = 512] [family = benchmark]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from trajtrack_mpcndqn_rlboost_amd import MpcConfig, BatchSolver, scenes
from trajtrack_mpcndqn_rlboost_amd.solver import variant_path

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
fam = sys.argv[3] if len(sys.argv) > 3 else "benchmark"
cfg = MpcConfig(N_hor=N)
sc = scenes.make_family(cfg, B, fam, seed=1234)
bs = BatchSolver(cfg, latency_batch=0, library=variant_path("trace"))
cap = 5200
bs.set_trace(cap)
res = bs.solve(sc["p"])
tr = bs.read_trace(B)
inner = res.num_inner_iterations
nls_all, lip_all = [], []
for b in range(B):
    n = min(int(inner[b]), cap)
    rows = tr[b, :n]
    ls = rows[:, 9]
    nls_all.append(ls[ls >= 0]); lip_all.append(rows[:, 7])
nls = np.concatenate(nls_all).astype(int); lip = np.concatenate(lip_all).astype(int)
print(f"N {N} B {B} family {fam}: {len(nls)} line searches, {len(lip)} steps, evals/solve {bs.last_eval_counts(B)[0].mean():.0f}")
h = np.bincount(nls, minlength=11) / len(nls)
print("halvings per line search (0 = tau 1 accepted):", " ".join(f"{i}:{x:.3f}" for i, x in enumerate(h)))
print("mean trials per line search", (nls + 1).mean())
hl = np.bincount(lip, minlength=4) / len(lip)
print("Lipschitz retries per step:", " ".join(f"{i}:{x:.4f}" for i, x in enumerate(hl[:6])))
# pairing (tau, tau/2) from the second trial on: evaluations saved / wasted
tr1 = nls + 1
pairs_from_second = np.where(tr1 == 1, 0, np.ceil((tr1 - 1) / 2))
wasted = np.where(tr1 == 1, 0, (2 * pairs_from_second) - (tr1 - 1))
print("pair from trial 2 on: paired evaluations per search", (2 * pairs_from_second).mean(), "of them wasted", wasted.mean())
pairs_all = np.ceil(tr1 / 2); wasted_all = 2 * pairs_all - tr1
print("pair from trial 1 on: paired evaluations per search", (2 * pairs_all).mean(), "of them wasted", wasted_all.mean())
