"""Child process of tests/test_dqn_learner.py::test_two_graph_update_with_the_rccl_all_reduce_on_the_gpu: the two-graph DQN update
around a flat RCCL all-reduce in a one-rank process group on cuda:0, compared with the eager update.  It runs in a process of its
own because an RCCL / HSA failure at process-group start ABORTS the interpreter (seen once on a pool box): that must fail one
test, not take the whole pytest session down.  Exit code 0 = equal; prints one line."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from trajtrack_mpcndqn_rlboost_amd import dqn_train  # noqa: E402


def main():
    dev = "cuda:0"
    torch.cuda.set_device(0)
    with socket.socket() as sk:                                   # a free port for the one-rank rendezvous
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(dev))
    try:
        def batch(seed, n=32):
            g = torch.Generator().manual_seed(seed)
            return {k: v.to(dev) for k, v in dict(obs=torch.rand(n, 46, generator=g) * 2 - 1, actions=torch.randint(0, 9, (n,), generator=g),
                                                   rewards=torch.randn(n, generator=g), next_obs=torch.rand(n, 46, generator=g) * 2 - 1,
                                                   dones=(torch.rand(n, generator=g) < 0.1).float()).items()}
        outs = []
        for collective in (False, True):
            torch.manual_seed(0)
            tr = dqn_train.DqnTrainer(device=dev, target_update_interval=3, force_collective=collective)
            if collective:
                tr.enable_graph(32)
                assert hasattr(tr, "_graph_b")                       # the two-graph structure was captured
            losses = [float((tr.update_graphed if collective else tr.update)(batch(20 + i))) for i in range(8)]
            outs.append((torch.cat([p.detach().reshape(-1) for p in tr.q_net.parameters()]).cpu(), losses))
        assert np.allclose(outs[0][1], outs[1][1], rtol=1e-5, atol=1e-7), (outs[0][1], outs[1][1])
        assert torch.allclose(outs[0][0], outs[1][0], rtol=1e-5, atol=1e-7)
        print(f"rccl one-rank two-graph update == eager update over 8 steps (backend {dist.get_backend()}, world {dist.get_world_size()})")
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
