"""TD3 in the loop with the batched HIP environment (BASELINE.json cfg5).

The reference trains one env through XML-RPC: 408 requests per episode, then `train(1000)`
(clientCFD.pde:35-85, server.py:148-170).  Here E envs of one GPU run the same episode protocol
in lock-step and the learner lives on the same device: observations, actions, records and the
replay buffer never leave HBM; at the end of an episode the E x 407 transitions (408 requests) are stored and
`train_steps` TD3 iterations run.

    python -m rlfluidcontrol_amd.train --envs 1024 --episodes 2 --train-steps 1000

Multi-GPU (cfg5: 8 x MI355X): one process per GPU over RCCL,
    python -m rlfluidcontrol_amd.train --gpus 8 --envs 1024        (spawns the ranks itself), or
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m rlfluidcontrol_amd.train ...
Every rank steps its shard and the one all-gather per agent step of `ShardedVecEnv` gives every rank the global
observation; every rank evaluates the same actor on it (identical weights, identical generator state), the reference's
only RL collective has the same shape (rank 0 asks the agent and `MPI_Bcast`s the two action values,
clientNektar/Fourier/src/drive.C:1561).  The LEARNER IS SHARDED: a rank stores the transitions of its own envs only
(E x 407 per episode) in a FIFO of its own that holds at least three whole episodes of all its envs (max(1 M, 3 x E x
410) transitions of 32 bytes: the reference's 1 M spans ~2400 single-env episodes, a 1 M / world share would keep only
the last ~120 agent steps of a 1024-env rank) -- a batch is world equal shares drawn from the ranks' own FIFOs instead
of one uniform draw over one FIFO (a deliberate deviation in sampling) -- and trains on a 1/world share of every batch (the
batch size must be a multiple of the world size); the gradients are averaged with one all-reduce per optimiser step
(~0.8 MB), so the weights stay bit-identical across ranks (tests/test_train_sharded_gloo.py): the ranks start from
rank 0's weights and generator state (broadcast at construction, seeded or not).
The rollout loop does not synchronise with the host per agent step: the number of agent steps of an episode is a
function of the config alone (float32 time stepping, `episode_plan`), rewards are accumulated on the device, and the
plan is checked against the device's `done` flags once, at the end.
"""
import argparse
import json
import os
import sys
import time

import torch

from . import td3
from .vec_env import HipBackend, ShardedVecEnv


def episode_plan(resolution, t_step=0.0075, init_time=1.0, end_time=50.0, substeps=16):
    """(agent-step calls, requests) of one episode started from t = 0 with a fresh averaging window, by the float32
    bookkeeping of k_agent_step / clientCFD.pde:35-56: a call runs sim-steps until a window completes, t reaches Time
    or the launch's step cap (afc_agent_step: warm-up steps + 2 windows + 8) is hit; a request follows every call whose
    window completed (408 requests)."""
    import numpy as np
    f = np.float32
    dt = f(f(t_step) * f(resolution))
    inc = f(dt / f(resolution))
    warm = int(f(init_time) / inc)
    cap = max(warm, 0) + 2 * substeps + 8  # afc_agent_step's max_steps
    t, cl, calls, requests = f(0), substeps, 0, 0
    while t < f(end_time):
        calls += 1
        done_window = False
        steps = 0
        while t < f(end_time) and steps < cap:
            t = f(t + inc)
            steps += 1
            if t > f(init_time):
                cl -= 1
                if cl <= 0:
                    cl = substeps
                    done_window = True
                    break
        if done_window:
            requests += 1
    return calls, requests


def run_episode(vec, learner, max_requests=None, stochastic=True, plan=None):
    """One pass of the reference's draw() loop for all envs: returns (requests, mean reward per request).
    plan = (calls, requests) from `episode_plan`: the loop then runs without a host synchronisation per agent step
    (all envs run the same fixed-dt episode in lock-step); without it the device's `done` flags are polled.  A plan the
    device disagrees with (envs not done after its last call: non-default time stepping the plan does not model) is
    not fatal: the episode is finished by polling and `run_episode.plan_failed` is set so the caller stops using it."""
    W, E = vec.world, vec.E
    action = torch.zeros(W * E, 2, device=vec.device)
    learner.start_episode()
    n = 0
    rew_sum = torch.zeros((), device=vec.device)
    run_episode.plan_failed = False
    if plan is not None:
        calls, requests = plan
        if max_requests is not None:
            requests = min(requests, max_requests)
        done = None
        for c in range(calls):
            obs, rew, done = vec.step(action)
            # a window that completes on the step that takes t to Time is still sent by the reference
            # (clientCFD.pde:36-56), one cut short by t >= Time is not: `requests` knows which it is
            if n >= requests:
                break
            action = learner.request_action(obs, stochastic=stochastic)
            if n > 0:
                rew_sum += rew.mean()
            n += 1
            if max_requests is not None and n >= max_requests:
                break
        if max_requests is not None or bool(done.all()):  # the one host synchronisation of the rollout
            return n, float(rew_sum) / max(n - 1, 1)
        run_episode.plan_failed = True  # finish the episode the slow way
    while True:
        obs, rew, done = vec.step(action)
        ended = bool(done.all())
        if ended and not vec.window_complete():
            break
        action = learner.request_action(obs, stochastic=stochastic)
        if n > 0:
            rew_sum += rew.mean()
        n += 1
        if ended or (max_requests is not None and n >= max_requests):
            break
    return n, float(rew_sum) / max(n - 1, 1)


def train(num_envs=1024, resolution=16, reynolds=100, episodes=1, train_steps=1000, device="cuda:0", seed=0,
          filter="None", explore_noise="Gaussian", init=None, max_requests=None, env_kwargs=None, batch_size=None,
          log=print):
    from . import build, env as envmod
    build.build()
    dev = torch.device(device)
    torch.cuda.set_device(dev)
    import torch.distributed as dist
    rank = dist.get_rank() if dist.is_initialized() else 0
    if rank != 0:
        log = (lambda *_a: None)  # one log line per episode, from rank 0
    g = envmod.AFCEnv(num_envs=num_envs, resolution=resolution, reynolds=reynolds, device=dev.index or 0,
                      **(env_kwargs or {}))
    if init is not None:
        g.set_reset_state(init["u"], init["v"], init["p"])
    vec = ShardedVecEnv(HipBackend(g, dev), num_envs, device=dev)
    if batch_size and batch_size % vec.world:
        raise ValueError("batch_size %d must be a multiple of the world size %d (sharded learner)" % (batch_size, vec.world))
    c = g.cfg
    # the sketch globals (callLearn, Cd, Cl) survive an episode roll-over in the reference's carry mode: the phase of the
    # averaging window then depends on history, and the loop polls the device instead of following a plan
    full_plan = episode_plan(c.resolution, c.t_step, c.init_time, c.end_time, c.substeps)
    plan = None if c.carry_quirk else full_plan
    # (the replay ring holds whole episodes of this env configuration: its requests per episode, not the reference case's)
    learner = td3.BatchedLearner(vec.world * num_envs, device=dev, filter=filter, explore_noise=explore_noise, seed=seed,
                                 shard=(vec.rank, vec.world), steps_per_episode=full_plan[1] + 2)
    if batch_size:
        learner.agent.bs = batch_size
    history = []
    for ep in range(episodes):
        g.reset()
        t0 = time.time()
        n, mean_r = run_episode(vec, learner, max_requests=max_requests, plan=plan)
        if plan is not None and run_episode.plan_failed:
            log("episode plan %r does not match the device (non-default time stepping?): polling from now on" % (plan,))
            plan = None
        torch.cuda.synchronize(dev)
        t1 = time.time()
        learner.train(train_steps)
        torch.cuda.synchronize(dev)
        t2 = time.time()
        rec = dict(episode=ep + 1, requests=n, mean_reward=mean_r, rollout_s=t1 - t0, train_s=t2 - t1,
                   buffer=learner.agent.replay_buffer.size, td3_iters=learner.agent.total_step_count,
                   world=vec.world, envs_total=vec.world * num_envs, collectives=vec.n_collectives,
                   grad_allreduces=learner.agent.n_grad_allreduce, replay_capacity=learner.agent.replay_capacity)
        history.append(rec)
        log(json.dumps(rec))
    g.close()
    return learner, history


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--envs", type=int, default=1024)
    ap.add_argument("--res", type=int, default=16)
    ap.add_argument("--reynolds", type=int, default=100)
    ap.add_argument("--episodes", type=int, default=1)
    ap.add_argument("--train-steps", type=int, default=1000)
    ap.add_argument("--max-requests", type=int, default=None)
    ap.add_argument("--filter", choices=["None", "Kalman"], default="None")
    ap.add_argument("--explore-noise", choices=["Gaussian", "Process"], default="Gaussian")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--gpus", type=int, default=1, help="ranks to spawn (one per GPU) when not started by a launcher")
    ap.add_argument("--init", default=None, help=".npz with u, v, p: the reset state (default: free stream)")
    a = ap.parse_args(argv)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # become the launcher BEFORE anything touches a GPU in this process
        import socket
        import subprocess
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), "-m", "rlfluidcontrol_amd.train"]
        raise SystemExit(subprocess.call(cmd + list(sys.argv[1:] if argv is None else argv)))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dev = "cuda:%d" % local_rank
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device(dev))
    init = None
    if a.init:
        import numpy as np
        init = np.load(a.init)
    try:
        train(num_envs=a.envs, resolution=a.res, reynolds=a.reynolds, episodes=a.episodes, train_steps=a.train_steps,
              max_requests=a.max_requests, filter=a.filter, explore_noise=a.explore_noise, seed=a.seed, device=dev,
              init=init)
    finally:
        if world > 1:
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
