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
