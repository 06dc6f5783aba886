"""N>1 path on CPU: two gloo processes, env-sharded ShardedVecEnv with a stub backend.
Checks the partition (global env g = rank*E + e), that every rank sees the same gathered
(obs, reward, done) and that exactly ONE collective runs per agent step."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rlfluidcontrol_amd.vec_env import ShardedVecEnv


class StubBackend:
    """obs = (global index, action sum), reward = -global index, done when index is odd"""

    def __init__(self, rank, E):
        self.rank, self.E = rank, E
        self.resets = []

    def agent_step_tensors(self, actions):
        g = torch.arange(self.E, dtype=torch.float32) + self.rank * self.E
        obs = torch.stack([g, actions.sum(dim=1)], dim=1)
        return obs, -g, (g.to(torch.int64) % 2).to(torch.uint8)

    def reset_mask(self, mask):
        self.resets.append(mask.clone())


def _worker(rank, world, port, E, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        be = StubBackend(rank, E)
        ve = ShardedVecEnv(be, E)
        actions = torch.arange(world * E * 2, dtype=torch.float32).reshape(world * E, 2)
        for _ in range(3):
            obs, rew, done = ve.step(actions)
        ve.reset_done(done)
        ok = (ve.n_collectives == 3 and obs.shape == (world * E, 2)
              and torch.equal(obs[:, 0], torch.arange(world * E, dtype=torch.float32))
              and torch.equal(obs[:, 1], actions.sum(dim=1))
              and torch.equal(rew, -torch.arange(world * E, dtype=torch.float32))
              and torch.equal(done, (torch.arange(world * E) % 2).bool())
              and torch.equal(be.resets[0], (torch.arange(E) + rank * E).to(torch.uint8) % 2))
        q.put((rank, bool(ok), ve.local_slice().start, ve.global_index(1)))
    finally:
        dist.destroy_process_group()


def test_sharded_vec_env_world2():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world, E = 2, 5
    procs = [ctx.Process(target=_worker, args=(r, world, port, E, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
    assert res == [(0, True, 0, 1), (1, True, 5, 6)]


def test_single_process_no_collective():
    be = StubBackend(0, 4)
    ve = ShardedVecEnv(be, 4)
    obs, rew, done = ve.step(torch.ones(4, 2))
    assert ve.n_collectives == 0 and torch.equal(obs[:, 1], torch.full((4,), 2.0))


class WakeStub:
    """a toy wake: obs depends on the env's own action history, so the sharding matters"""

    def __init__(self, rank, E):
        self.g = torch.arange(E, dtype=torch.float32) + rank * E
        self.state = torch.zeros(E, 2)
        self.k = 0

    def agent_step_tensors(self, actions):
        self.k += 1
        self.state = 0.9 * self.state + 0.1 * actions + 0.01 * torch.sin(self.g + self.k)[:, None]
        obs = torch.stack([self.state[:, 0] - self.state[:, 1], 1.0 + self.state.sum(1)], 1)
        done = torch.full((len(self.g),), 1 if self.k >= 7 else 0, dtype=torch.uint8)
        return obs, -obs[:, 1], done

    def reset_mask(self, mask):
        pass


def _learner_worker(rank, world, port, E, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rlfluidcontrol_amd import td3, train
        ve = ShardedVecEnv(WakeStub(rank, E), E)
        learner = td3.BatchedLearner(world * E, seed=11)  # replicated: same seed on every rank
        learner.agent.bs = 8
        n, _ = train.run_episode(ve, learner)
        learner.train(5)
        w = torch.cat([p.detach().reshape(-1) for p in learner.agent.main_actor.parameters()])
        q.put((rank, n, ve.n_collectives, learner.agent.replay_buffer.size, float(w.double().sum()),
               float(torch.stack(learner.action_record).double().sum())))
    finally:
        dist.destroy_process_group()


def test_replicated_learner_stays_identical_across_ranks():
    """no gradient collective -- every rank sees the gathered obs and runs the same learner"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world, E = 2, 3
    procs = [ctx.Process(target=_learner_worker, args=(r, world, port, E, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(60)
    (r0, n0, c0, b0, w0, a0), (r1, n1, c1, b1, w1, a1) = res
    assert (n0, c0, b0) == (n1, c1, b1) == (6, 7, 5 * world * E)  # 6 requests, 7 gathers, 5 transitions per env
    assert w0 == w1 and a0 == a1  # bit-identical weights and actions on both ranks
