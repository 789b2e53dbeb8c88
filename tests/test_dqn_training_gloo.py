"""CPU, world_size 2 over gloo: the data-parallel DQN update (flat-bucket gradient all-reduce) gives the same
parameters as a single process on the whole batch."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _make_batch(n, seed):
    g = torch.Generator().manual_seed(seed)
    return dict(obs=torch.rand(n, 46, generator=g) * 2 - 1, actions=torch.randint(0, 9, (n,), generator=g),
                rewards=torch.randn(n, generator=g), next_obs=torch.rand(n, 46, generator=g) * 2 - 1,
                dones=(torch.rand(n, generator=g) < 0.1).float())


def _run_updates(trainer, batches, lo=None, hi=None):
    for b in batches:
        shard = {k: v[lo:hi] for k, v in b.items()} if lo is not None else b
        trainer.update(shard)
    return torch.cat([p.detach().reshape(-1) for p in trainer.q_net.parameters()])


def _worker(rank, world, port, tmpdir, double_q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from trajtrack_mpcndqn_rlboost_amd.dqn import QNetwork
    from trajtrack_mpcndqn_rlboost_amd.dqn_train import DqnTrainer
    torch.manual_seed(rank)            # ranks construct DIFFERENT networks: the trainer broadcasts rank 0's weights
    net = QNetwork()
    tr = DqnTrainer(net, double_q=double_q, target_update_interval=2)
    assert tr.world == world
    tr.assert_replicas_equal()
    assert all(torch.equal(a, b) for a, b in zip(tr.q_net.parameters(), tr.q_net_target.parameters()))
    batches = [_make_batch(32, 100 + i) for i in range(5)]
    per = 32 // world
    flat = _run_updates(tr, batches, rank * per, (rank + 1) * per)
    tr.assert_replicas_equal()
    np.save(os.path.join(tmpdir, f"w_{rank}.npy"), flat.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _single(double_q):
    sys.path.insert(0, ROOT)
    from trajtrack_mpcndqn_rlboost_amd.dqn import QNetwork
    from trajtrack_mpcndqn_rlboost_amd.dqn_train import DqnTrainer
    torch.manual_seed(0)
    tr = DqnTrainer(QNetwork(), double_q=double_q, target_update_interval=2)
    return _run_updates(tr, [_make_batch(32, 100 + i) for i in range(5)]).numpy()


def test_two_rank_update_equals_single_process_update(tmp_path):
    for k, double_q in enumerate((False, True)):
        d = tmp_path / f"case{k}"
        d.mkdir()
        port = 29700 + (os.getpid() % 1000) + k
        mp.start_processes(_worker, args=(2, port, str(d), double_q), nprocs=2, join=True, start_method="spawn")
        w0, w1 = np.load(d / "w_0.npy"), np.load(d / "w_1.npy")
        assert np.array_equal(w0, w1)                              # replicas stay identical
        assert np.max(np.abs(w0 - _single(double_q))) < 2e-6       # and equal the whole-batch update (fp32)


def test_reference_hyper_parameters_and_target_sync():
    sys.path.insert(0, ROOT)
    from trajtrack_mpcndqn_rlboost_amd.dqn_train import DqnTrainer
    tr = DqnTrainer(target_update_interval=3)
    assert tr.gamma == 0.98 and tr.max_grad_norm == 10.0 and not tr.double_q
    assert tr.optimizer.param_groups[0]["lr"] == 1e-4
    b = _make_batch(32, 7)
    before = [p.clone() for p in tr.q_net_target.parameters()]
    tr.update(b); tr.update(b)
    assert all(torch.equal(a, c) for a, c in zip(before, tr.q_net_target.parameters()))
    tr.update(b)                                                    # third update: target <- online
    assert all(torch.equal(a, c) for a, c in zip(tr.q_net.parameters(), tr.q_net_target.parameters()))
    assert tr._bucket.numel() == 1177
