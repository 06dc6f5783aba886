"""The sharded learner of train.py on CPU: two gloo ranks, stub envs.  Every rank acts for all envs on the gathered
observations, stores the transitions of ITS OWN envs only, trains on half of every batch with gradients averaged by one
all-reduce per optimiser step -- and the weights must stay bit-identical across the ranks.  Also: the rollout follows
the episode plan without polling the device, and the plan (float32 time stepping) gives the reference's 408 requests."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rlfluidcontrol_amd import td3
from rlfluidcontrol_amd.train import episode_plan, run_episode
from rlfluidcontrol_amd.vec_env import ShardedVecEnv


def test_episode_plan_matches_the_reference_count():
    # 6 667 sim-steps, first request after step 149, then every 16: 408 requests, for
    # every resolution; one more call runs the episode out without completing a window
    for res in (8, 16, 24, 32):
        assert episode_plan(res) == (409, 408)


class StubBackend:
    """obs depends on the env's global index and the action; done on the call the plan says is the last"""

    def __init__(self, rank, E, last_call):
        self.rank, self.E, self.last_call, self.calls = rank, E, last_call, 0

    def agent_step_tensors(self, actions):
        self.calls += 1
        g = torch.arange(self.E, dtype=torch.float32) + self.rank * self.E
        obs = torch.stack([0.01 * g + 0.1 * actions[:, 0], 1.0 + 0.05 * actions[:, 1] - 0.001 * self.calls], dim=1)
        done = torch.full((self.E,), int(self.calls >= self.last_call), dtype=torch.uint8)
        return obs, -obs[:, 1], done

    def reset_mask(self, mask):
        pass


def _worker(rank, world, port, E, T, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        vec = ShardedVecEnv(StubBackend(rank, E, T + 1), E)
        learner = td3.BatchedLearner(world * E, device="cpu", seed=3, shard=(rank, world))
        learner.agent.bs = 16
        n, mean_r = run_episode(vec, learner, plan=(T + 1, T))
        learner.train(12)
        flat = torch.cat([p.detach().reshape(-1) for net in (learner.agent.main_actor, learner.agent.main_critic_1,
                                                              learner.agent.main_critic_2, learner.agent.target_actor)
                          for p in net.parameters()])
        both = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(both, flat)
        q.put((rank, n, learner.agent.replay_buffer.size, bool(torch.equal(both[0], both[1])),
               learner.agent.n_grad_allreduce, vec.n_collectives, float(learner.agent.replay_buffer.obs1_buf[0, 0]),
               learner.agent.total_step_count))
    finally:
        dist.destroy_process_group()


def test_sharded_learner_world2_keeps_weights_identical():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world, E, T = 2, 6, 9
    procs = [ctx.Process(target=_worker, args=(r, world, port, E, T, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, n, size, same, nred, ncoll, first_obs, iters in res:
        assert n == T                      # requests follow the plan, no polling
        assert size == E * (T - 1)         # this rank's envs only: E x (T-1) transitions, not world x that
        assert same                        # weights bit-identical across ranks after 12 sharded iterations
        assert iters == 12 and nred == 12 + 6  # one all-reduce per critic step + one per (delayed) actor step
        assert ncoll == T + 1              # one gather per agent-step call
    # rank r stored the env block [r*E, (r+1)*E): its first stored observation is that of global env r*E
    assert abs(res[0][6] - 0.0) < 1e-6 and abs(res[1][6] - 0.01 * E) < 1e-6
