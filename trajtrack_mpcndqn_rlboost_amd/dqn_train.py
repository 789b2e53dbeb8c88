"""Data-parallel DQN update for the action-value head (BASELINE.json config 5).

Hyper-parameters are the reference's SB3 settings (``src/test_block_rl.py:77-86`` and the ``data`` member of
``Model/ray/best_model.zip``): Huber loss, Adam lr 1e-4, gamma 0.98, batch 32, target sync every 10 000 steps,
gradient clip 10.  The reference trains a vanilla DQN in one process; this build adds (a) an optional Double-DQN
target (``double_q``; off by default = reference semantics) and (b) data parallelism: every rank computes the
gradient of its shard of the batch, ONE all-reduce of a single flat bucket of 1 177 fp32 values sums them
(RCCL over xGMI on MI355X -- 4.7 KB, pure latency), then every rank applies the same optimiser step.
"""
from __future__ import annotations

import copy
from typing import Callable, Dict, Optional

import torch
import torch.distributed as dist
from torch import nn
from torch.nn import functional as F

from .dqn import QNetwork


class DqnTrainer:
    def __init__(self, q_net: Optional[QNetwork] = None, lr: float = 1e-4, gamma: float = 0.98,
                 target_update_interval: int = 10_000, max_grad_norm: float = 10.0, double_q: bool = False,
                 device: str = "cpu", force_collective: bool = False):
        """``force_collective``: run the gradient all-reduce (and the two-graph replay built around it) even when the process
        group has ONE rank -- the multi-rank code path, collective included, on a single GPU (tests)."""
        self.q_net = (q_net if q_net is not None else QNetwork()).to(device)
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        if force_collective and not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("force_collective needs an initialised process group")
        self._collective = self.world > 1 or force_collective
        self._params = [p for p in self.q_net.parameters()]
        self._bucket = torch.zeros(sum(p.numel() for p in self._params), dtype=torch.float32, device=device)
        if self.world > 1:
            # Replicas must START equal: the same averaged gradient applied to different weights never re-converges.
            # Rank 0's weights win (whatever seeding or checkpoint loading happened before on the other ranks).
            with torch.no_grad():
                self._pack([p.detach() for p in self._params])
                dist.broadcast(self._bucket, src=0)
                self._unpack([p for p in self._params])
        self.q_net_target = copy.deepcopy(self.q_net).requires_grad_(False)
        self.optimizer = torch.optim.Adam(self.q_net.parameters(), lr=lr)
        self.gamma, self.max_grad_norm, self.double_q = gamma, max_grad_norm, double_q
        self.target_update_interval = target_update_interval
        self.num_updates = 0

    def td_target(self, rewards, next_obs, dones) -> torch.Tensor:
        with torch.no_grad():
            next_q = self.q_net_target(next_obs)
            if self.double_q:      # action chosen by the online net, valued by the target net
                best = self.q_net(next_obs).argmax(dim=1, keepdim=True)
                next_v = next_q.gather(1, best).squeeze(1)
            else:                  # vanilla DQN (the reference)
                next_v = next_q.max(dim=1).values
            return rewards + (1.0 - dones) * self.gamma * next_v

    def _pack(self, tensors):
        off = 0
        for t in tensors:
            n = t.numel()
            self._bucket[off:off + n].copy_(t.reshape(-1))
            off += n

    def _unpack(self, tensors, scale: Optional[float] = None):
        off = 0
        for t in tensors:
            n = t.numel()
            src = self._bucket[off:off + n].view_as(t)
            t.copy_(src if scale is None else src * scale)
            off += n

    def _all_reduce_gradients(self):
        """One flat bucket: pack, sum over ranks, average, unpack."""
        self._pack([p.grad for p in self._params])
        dist.all_reduce(self._bucket, op=dist.ReduceOp.SUM)
        self._unpack([p.grad for p in self._params], 1.0 / self.world)

    def assert_replicas_equal(self) -> None:
        """Debug aid: every rank holds bit-identical online weights (max - min over ranks of the flat bucket is 0)."""
        if self.world == 1:
            return
        with torch.no_grad():
            self._pack([p.detach() for p in self._params])
            hi, lo = self._bucket.clone(), self._bucket.clone()
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            if not torch.equal(hi, lo):
                raise AssertionError(f"Q-network replicas differ across ranks (max |diff| {float((hi - lo).abs().max()):.3e})")

    def update(self, batch: Dict[str, torch.Tensor]) -> float:
        """batch (this rank's shard): obs [b,46] f32, actions [b] i64, rewards [b], next_obs [b,46], dones [b]."""
        loss = self._eager_step(batch)
        self.num_updates += 1
        if self.target_update_interval and self.num_updates % self.target_update_interval == 0:
            self.sync_target()
        return loss

    # ---- hipGraph path: the update is ~40 tiny kernels (MLP forward / backward, Huber loss, clip, Adam) on 1 177
    # parameters, i.e. pure launch latency; captured once, it replays as ONE graph launch
    def enable_graph(self, batch_size: int, obs_dim: int = 46) -> None:
        """Capture ``update`` for a fixed batch size into HIP graphs (``torch.cuda.CUDAGraph`` is hipGraph on ROCm).
        One process: ONE graph.  Several ranks: TWO graphs with the flat-bucket all-reduce between them -- graph A =
        forward, backward and packing the gradients into the bucket; the collective (RCCL) is enqueued eagerly on the same
        stream; graph B = averaging, unpacking, clipping and the Adam step.  Every rank must call this (the warm-up steps
        contain collectives)."""
        dev = self._bucket.device
        if dev.type != "cuda":
            raise RuntimeError("enable_graph needs the GPU")
        lr = self.optimizer.param_groups[0]["lr"]
        self.optimizer = torch.optim.Adam(self.q_net.parameters(), lr=lr, capturable=True)
        self._g_batch = dict(obs=torch.zeros(batch_size, obs_dim, device=dev), actions=torch.zeros(batch_size, dtype=torch.int64, device=dev),
                             rewards=torch.zeros(batch_size, device=dev), next_obs=torch.zeros(batch_size, obs_dim, device=dev),
                             dones=torch.zeros(batch_size, device=dev))
        self._g_loss = torch.zeros((), device=dev)
        keep_interval, self.target_update_interval = self.target_update_interval, 0
        state = ([p.detach().clone() for p in self._params], self.num_updates)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # warm-up off the capture (allocations, Adam state)
            for _ in range(3):
                self._eager_step(self._g_batch)
        torch.cuda.current_stream().wait_stream(side)
        self._graph = torch.cuda.CUDAGraph()
        if not self._collective:
            with torch.cuda.graph(self._graph):
                self._g_loss.copy_(self._eager_step(self._g_batch))
        else:
            with torch.cuda.graph(self._graph):
                self._g_loss.copy_(self._backward(self._g_batch))
                self._pack([p.grad for p in self._params])
            self._graph_b = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph_b, pool=self._graph.pool()):
                self._unpack([p.grad for p in self._params], 1.0 / self.world)
                self._apply()
        with torch.no_grad():                              # undo the warm-up / capture steps on the zero batch
            for p, p0 in zip(self._params, state[0]):
                p.copy_(p0)
            for st in self.optimizer.state.values():
                st["exp_avg"].zero_(); st["exp_avg_sq"].zero_(); st["step"].zero_()
        self.num_updates = state[1]
        self.target_update_interval = keep_interval

    def _backward(self, batch: Dict[str, torch.Tensor]) -> torch.Tensor:
        target = self.td_target(batch["rewards"], batch["next_obs"], batch["dones"])
        q = self.q_net(batch["obs"]).gather(1, batch["actions"].view(-1, 1)).squeeze(1)
        loss = F.smooth_l1_loss(q, target)
        self.optimizer.zero_grad(set_to_none=False)
        loss.backward()
        return loss.detach()

    def _apply(self) -> None:
        nn.utils.clip_grad_norm_(self._params, self.max_grad_norm)
        self.optimizer.step()

    def _eager_step(self, batch: Dict[str, torch.Tensor]) -> torch.Tensor:
        loss = self._backward(batch)
        if self._collective:
            self._all_reduce_gradients()
        self._apply()
        return loss

    def update_graphed(self, batch: Dict[str, torch.Tensor]) -> torch.Tensor:
        """Same arithmetic as :meth:`update`, replayed from the captured graph(s) (batch size fixed by enable_graph)."""
        for k, v in self._g_batch.items():
            v.copy_(batch[k])
        self._graph.replay()
        if self._collective:
            dist.all_reduce(self._bucket, op=dist.ReduceOp.SUM)
            self._graph_b.replay()
        self.num_updates += 1
        if self.target_update_interval and self.num_updates % self.target_update_interval == 0:
            self.sync_target()
        return self._g_loss

    def load_optimizer_state(self, sd: Dict) -> None:
        """Restore Adam's moments and step counts.  Once ``enable_graph`` has captured the update, the graph is bound to the
        optimiser's state TENSORS: ``Optimizer.load_state_dict`` would replace them with new ones that the replayed graph never
        sees (the restored moments would be ignored and later checkpoints would save the stale copies), so the loaded values
        are copied IN PLACE into the tensors the graph updates."""
        if not hasattr(self, "_graph"):
            self.optimizer.load_state_dict(sd)
            return
        params = [p for g in self.optimizer.param_groups for p in g["params"]]
        ids = [i for g in sd["param_groups"] for i in g["params"]]
        if len(ids) != len(params):
            raise ValueError(f"optimizer checkpoint holds {len(ids)} parameters, this optimizer {len(params)}")
        # everything the captured graph has baked in is compared BEFORE anything is copied: a checkpoint that does not fit must
        # leave the live optimiser state as it was
        if len(sd["param_groups"]) != len(self.optimizer.param_groups):
            raise ValueError(f"optimizer checkpoint holds {len(sd['param_groups'])} parameter groups, this optimizer {len(self.optimizer.param_groups)}")
        for g, gs in zip(self.optimizer.param_groups, sd["param_groups"]):
            for key in ("lr", "betas", "eps", "weight_decay", "amsgrad", "maximize"):
                as_tuple = lambda v: tuple(v) if isinstance(v, (tuple, list)) else (v,)
                if key in g and key in gs and as_tuple(g[key]) != as_tuple(gs[key]):
                    raise ValueError(f"the captured graph holds the optimiser's {key} = {g[key]!r} it was captured with; the checkpoint has "
                                     f"{gs[key]!r} (load it before enable_graph, or capture again)")
        for p, i in zip(params, ids):
            src = sd["state"].get(i)
            if src is not None:
                for k in ("exp_avg", "exp_avg_sq"):
                    if tuple(torch.as_tensor(src[k]).shape) != tuple(self.optimizer.state[p][k].shape):
                        raise ValueError(f"optimizer checkpoint: {k} of parameter {i} has shape {tuple(torch.as_tensor(src[k]).shape)}, "
                                         f"expected {tuple(self.optimizer.state[p][k].shape)}")
        with torch.no_grad():
            for p, i in zip(params, ids):
                st, src = self.optimizer.state[p], sd["state"].get(i)
                for k in ("exp_avg", "exp_avg_sq", "step"):
                    if src is None:
                        st[k].zero_()
                    else:
                        st[k].copy_(torch.as_tensor(src[k]).to(device=st[k].device, dtype=st[k].dtype))

    def sync_target(self) -> None:
        """Hard target update (SB3 ``polyak_update`` with tau = 1)."""
        self.q_net_target.load_state_dict(self.q_net.state_dict())
        self.num_target_syncs = getattr(self, "num_target_syncs", 0) + 1


# ----------------------------------------------------------------------------------------------------------------------
# collection + learning loop over the batched environment (SB3 ``DQN.learn`` semantics of the reference's training
# script: src/test_block_rl.py:68-86 -- n_envs vectorised environments, learning_starts 50 000, train_freq 4 steps,
# gradient_steps -1 (one gradient step per collected transition), batch 32, buffer 1e6, epsilon 1.0 -> 0.05 over the
# first 20 % of the run, hard target update every 10 000 environment steps)
# ----------------------------------------------------------------------------------------------------------------------
def flatten_observation(obs: Dict[str, torch.Tensor]) -> torch.Tensor:
    """SB3's CombinedExtractor concatenates the dict entries in sorted key order: ``external`` (32) then ``internal``
    (14) -- the input layout of the reference's trained network (see dqn.py)."""
    return torch.cat([obs["external"], obs["internal"]], dim=1)


class ReplayBuffer:
    """Uniform ring buffer, resident on the environment's device (transitions never visit the host)."""

    def __init__(self, capacity: int, obs_dim: int, device):
        self.capacity, self.device = int(capacity), device
        self.obs = torch.zeros(self.capacity, obs_dim, dtype=torch.float32, device=device)
        self.next_obs = torch.zeros_like(self.obs)
        self.actions = torch.zeros(self.capacity, dtype=torch.int64, device=device)
        self.rewards = torch.zeros(self.capacity, dtype=torch.float32, device=device)
        self.dones = torch.zeros(self.capacity, dtype=torch.float32, device=device)
        self.pos, self.size = 0, 0

    def add(self, obs, next_obs, actions, rewards, dones) -> None:
        n = obs.shape[0]
        idx = (self.pos + torch.arange(n, device=self.device)) % self.capacity
        self.obs[idx], self.next_obs[idx] = obs, next_obs
        self.actions[idx] = actions.to(torch.int64)
        self.rewards[idx] = rewards.to(torch.float32)
        self.dones[idx] = dones.to(torch.float32)
        self.pos = (self.pos + n) % self.capacity
        self.size = min(self.size + n, self.capacity)

    def sample(self, n: int, generator: Optional[torch.Generator] = None) -> Dict[str, torch.Tensor]:
        idx = torch.randint(0, self.size, (n,), device=self.device, generator=generator)
        return dict(obs=self.obs[idx], actions=self.actions[idx], rewards=self.rewards[idx],
                    next_obs=self.next_obs[idx], dones=self.dones[idx])


def exploration_rate(step: int, total: int, fraction: float = 0.2, initial: float = 1.0, final: float = 0.05) -> float:
    """SB3 ``get_linear_fn``: linear from ``initial`` to ``final`` over the first ``fraction`` of the run."""
    progress = step / float(total)
    if progress > fraction:
        return final
    return initial + progress * (final - initial) / fraction


class DqnLearner:
    """Collect with an epsilon-greedy policy on a batched environment and update the Q-network.

    ``env`` needs ``B``, ``device``, ``reset()`` and ``step(actions, auto_reset=True)`` with the contract of
    :class:`rl_env.BatchedRaysEnv`.  With ``torch.distributed`` initialised every rank drives its own shard of
    environments and its own buffer; gradients are summed by :class:`DqnTrainer`'s single flat all-reduce."""

    def __init__(self, env, trainer: Optional[DqnTrainer] = None, buffer_size: int = 1_000_000,
                 learning_starts: int = 50_000, batch_size: int = 32, train_freq: int = 4, gradient_steps: int = -1,
                 target_update_interval: int = 10_000, exploration_fraction: float = 0.2,
                 exploration_initial_eps: float = 1.0, exploration_final_eps: float = 0.05, seed: int = 0,
                 use_graph: bool = False, track_episodes: bool = True):
        self.env = env
        self.device = env.device
        self.trainer = trainer if trainer is not None else DqnTrainer(device=str(self.device))
        self.trainer.target_update_interval = 0   # the learner syncs on environment steps, as SB3 does
        self.buffer = ReplayBuffer(buffer_size, 46, self.device)
        self.learning_starts, self.batch_size, self.train_freq = learning_starts, batch_size, train_freq
        self.gradient_steps, self.target_update_interval = gradient_steps, target_update_interval
        self.eps = (exploration_fraction, exploration_initial_eps, exploration_final_eps)
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(seed + (dist.get_rank() if dist.is_available() and dist.is_initialized() else 0))
        self.num_timesteps, self.n_calls = 0, 0
        self.episode_returns, self.episode_successes = [], []
        # track_episodes=False keeps the episode statistics as device counters (no host synchronisation in the loop)
        self.track_episodes = track_episodes
        self.ep_count = torch.zeros((), dtype=torch.float64, device=self.device)
        self.ep_return_sum = torch.zeros((), dtype=torch.float64, device=self.device)
        self.ep_success_sum = torch.zeros((), dtype=torch.float64, device=self.device)
        self.use_graph = use_graph
        if use_graph:
            self.trainer.enable_graph(batch_size)
        # state of the collection loop between two learn() calls (a run can be checkpointed and resumed in the middle)
        self._obs, self._ep_return, self.n_updates = None, None, 0
        self._last_loss = torch.zeros((), device=self.device)

    def act(self, obs_flat: torch.Tensor, epsilon: float) -> torch.Tensor:
        greedy = self.trainer.q_net.greedy_actions(obs_flat)
        B = obs_flat.shape[0]
        explore = torch.rand(B, device=self.device, generator=self.gen) < epsilon
        rand = torch.randint(0, 9, (B,), device=self.device, generator=self.gen)
        return torch.where(explore, rand, greedy)

    # ---- checkpoint / resume: what SB3's model.save / EvalCallback(best_model_save_path) / Algorithm.load do for the
    # reference (src/test_block_rl.py:73-76,89-96), extended by everything an EXACT continuation needs -- optimiser moments,
    # counters, random generator, replay buffer, the environments' own state and the observation the loop stands at.
    def state_dict(self, include_buffer: bool = True) -> Dict:
        tr = self.trainer
        d = dict(q_net=tr.q_net.state_dict(), q_net_target=tr.q_net_target.state_dict(), optimizer=tr.optimizer.state_dict(),
                 num_updates=tr.num_updates, num_target_syncs=getattr(tr, "num_target_syncs", 0),
                 num_timesteps=self.num_timesteps, n_calls=self.n_calls, n_updates=self.n_updates,
                 generator=self.gen.get_state(), episode_returns=list(self.episode_returns),
                 episode_successes=list(self.episode_successes), ep_count=self.ep_count.clone(),
                 ep_return_sum=self.ep_return_sum.clone(), ep_success_sum=self.ep_success_sum.clone(),
                 obs=None if self._obs is None else self._obs.clone(),
                 ep_return=None if self._ep_return is None else self._ep_return.clone())
        if include_buffer:
            b, n = self.buffer, self.buffer.size
            d["buffer"] = dict(pos=b.pos, size=n, obs=b.obs[:n].clone(), next_obs=b.next_obs[:n].clone(),
                               actions=b.actions[:n].clone(), rewards=b.rewards[:n].clone(), dones=b.dones[:n].clone())
        if hasattr(self.env, "state_dict"):
            d["env"] = self.env.state_dict()
        return d

    def load_state_dict(self, d: Dict) -> None:
        tr = self.trainer
        tr.q_net.load_state_dict(d["q_net"]); tr.q_net_target.load_state_dict(d["q_net_target"])
        tr.load_optimizer_state(d["optimizer"])     # in place when the update has been captured into a graph
        tr.num_updates, tr.num_target_syncs = d["num_updates"], d["num_target_syncs"]
        self.num_timesteps, self.n_calls, self.n_updates = d["num_timesteps"], d["n_calls"], d["n_updates"]
        self.gen.set_state(d["generator"].cpu())   # a generator state is a host ByteTensor, also for a device generator
        self.episode_returns, self.episode_successes = list(d["episode_returns"]), list(d["episode_successes"])
        self.ep_count.copy_(d["ep_count"]); self.ep_return_sum.copy_(d["ep_return_sum"]); self.ep_success_sum.copy_(d["ep_success_sum"])
        self._obs = None if d["obs"] is None else d["obs"].to(self.device)
        self._ep_return = None if d["ep_return"] is None else d["ep_return"].to(self.device)
        if "buffer" in d:
            b, src = self.buffer, d["buffer"]
            n = src["size"]
            if n > b.capacity:
                raise ValueError(f"checkpointed buffer holds {n} transitions, this buffer only {b.capacity}")
            b.obs[:n], b.next_obs[:n] = src["obs"].to(self.device), src["next_obs"].to(self.device)
            b.actions[:n], b.rewards[:n], b.dones[:n] = (src[k].to(self.device) for k in ("actions", "rewards", "dones"))
            b.pos, b.size = src["pos"] % b.capacity, n
        if "env" in d and hasattr(self.env, "load_state_dict"):
            self.env.load_state_dict(d["env"])

    def save(self, path: str, include_buffer: bool = True) -> None:
        torch.save(self.state_dict(include_buffer), path)

    def load(self, path: str) -> None:
        # tensors, numbers, lists and dicts only (the generator state is a ByteTensor): no pickled code is executed
        self.load_state_dict(torch.load(path, map_location=self.device, weights_only=True))

    def save_model(self, path: str) -> None:
        """The policy alone (the reference's final_model / best_model): the Q-network's weights."""
        torch.save(self.trainer.q_net.state_dict(), path)

    def learn(self, total_timesteps: int, callback: Optional[Callable[["DqnLearner"], None]] = None,
              stop_at: Optional[int] = None) -> Dict[str, float]:
        """Collect and learn until ``total_timesteps`` (the horizon of the exploration schedule) -- or until ``stop_at``, a
        point to checkpoint at; a later call, also on a learner restored with ``load``, continues exactly there."""
        env, B = self.env, self.env.B
        if self._obs is None:
            self._obs = flatten_observation(env.reset())
            self._ep_return = torch.zeros(B, dtype=torch.float64, device=self.device)
        obs, ep_return = self._obs, self._ep_return
        last_loss = self._last_loss
        n_updates = self.n_updates
        end = total_timesteps if stop_at is None else min(stop_at, total_timesteps)
        while self.num_timesteps < end:
            collected = 0
            for _ in range(self.train_freq):
                eps = exploration_rate(self.num_timesteps, total_timesteps, *self.eps)
                actions = self.act(obs, eps)
                nxt, reward, terminated, truncated, info = env.step(actions, auto_reset=True)
                done = terminated | truncated
                nxt_flat = flatten_observation(nxt)
                # the transition stores the observation the episode ended in, not the first one of the next episode;
                # a time-limit truncation is not a terminal state for the bootstrap (SB3 handle_timeout_termination)
                stored_next = flatten_observation(info["terminal_observation"]) if "terminal_observation" in info else nxt_flat
                self.buffer.add(obs, stored_next, actions, reward, terminated)
                ep_return += reward
                if self.track_episodes:
                    if bool(done.any()):
                        self.episode_returns += ep_return[done].tolist()
                        self.episode_successes += info["success"][done].tolist()
                else:
                    self.ep_count += done.sum()
                    self.ep_return_sum += (ep_return * done).sum()
                    self.ep_success_sum += (info["success"] & done).sum()
                ep_return = torch.where(done, torch.zeros_like(ep_return), ep_return)
                obs = nxt_flat
                self.num_timesteps += B
                self.n_calls += 1
                collected += B
                if self.n_calls % max(self.target_update_interval // B, 1) == 0:
                    self.trainer.sync_target()
                if self.num_timesteps >= total_timesteps:
                    break
            if self.num_timesteps > self.learning_starts and self.buffer.size >= self.batch_size:
                steps = self.gradient_steps if self.gradient_steps >= 0 else collected
                for _ in range(steps):
                    step = self.trainer.update_graphed if self.use_graph else self.trainer.update
                    last_loss = step(self.buffer.sample(self.batch_size, self.gen))
                n_updates += steps
            self._obs, self._ep_return, self.n_updates, self._last_loss = obs, ep_return, n_updates, last_loss
            if callback is not None:
                callback(self)
        self._obs, self._ep_return, self.n_updates, self._last_loss = obs, ep_return, n_updates, last_loss
        if not self.track_episodes:       # whole-run averages from the device counters
            n = float(self.ep_count)
            return dict(timesteps=self.num_timesteps, updates=n_updates, loss=float(last_loss), episodes=int(n),
                        mean_return=float(self.ep_return_sum) / n if n else float("nan"),
                        success_rate=float(self.ep_success_sum) / n if n else float("nan"))
        recent = self.episode_returns[-100:]
        return dict(timesteps=self.num_timesteps, updates=n_updates, loss=float(last_loss),
                    episodes=len(self.episode_returns),
                    mean_return=float(sum(recent) / len(recent)) if recent else float("nan"),
                    success_rate=float(sum(self.episode_successes[-100:]) / max(1, len(self.episode_successes[-100:]))))
