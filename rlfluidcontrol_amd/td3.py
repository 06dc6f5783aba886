"""TD3 learner next to the batched envs -- counterpart of the reference's
TF-1.x agent (`server/agent_TD3.py`, `server/utils.py`, `server/server.py:61-65,148-170`).

Same algorithm and constants; re-designed for the batched setting: every buffer is a device tensor,
a whole batch of envs is filtered / recorded / acted on per call, and nothing crosses to the host
inside the agent-step loop.  PyTorch is the plumbing here (autograd + Adam on 0.2 M parameters);
the hot path is the HIP environment.

Reference constants (agent_TD3.py:12-27): lr 1e-4 (Adam, TF defaults), gamma 0.99, tau 0.005,
batch 512, buffer 1e6, policy delay 2, exploration sigma 0.1, target smoothing sigma 0.2 clipped to
0.5, actor 2->256->256->2 (ReLU, ReLU, tanh), twin critics (2+2)->256->256->1, Glorot-uniform
kernels and zero biases (tf.layers.dense defaults).
"""
import math

import torch
import torch.nn as nn

REWARD_ACTION_COST = math.pi * (1 / 8) * 0.0097 * (3.66 ** 3)  # server.py:64 (= 0.186756...)


def reward_func(next_state, action):
    """server.py:61-65: r = -Cd(next) - pi/8 * 0.0097 * 3.66^3 * sum |a|^3.
    next_state [...,2] = (Cl, Cd); action [...,A]."""
    return -next_state[..., 1] - REWARD_ACTION_COST * (action.abs() ** 3).sum(-1)


class KalmanFilter:
    """utils.py:3-23: scalar-gain filter applied to every observation component, one independent
    copy per env (the reference keeps one for its single env)."""

    def __init__(self, Q=0.25, R=11.0):
        self.Q, self.R = Q, R
        self.posteri = None
        self.posteri_error = 1.0

    def reset(self, mask=None):
        if mask is None or self.posteri is None:
            self.posteri = None
            self.posteri_error = 1.0
        else:  # per-env reset
            self.posteri = torch.where(mask[:, None], torch.zeros_like(self.posteri), self.posteri)
            self.posteri_error = torch.where(mask[:, None], torch.ones_like(self.posteri_error), self.posteri_error)

    def estimate(self, measurement=None):
        if measurement is not None:
            if self.posteri is None:
                self.posteri = torch.zeros_like(measurement)
                self.posteri_error = torch.ones_like(measurement)
            priori = self.posteri
            priori_error = self.posteri_error + self.Q
            K = priori_error / (priori_error + self.R)
            self.posteri = priori + K * (measurement - priori)
            self.posteri_error = (1 - K) * priori_error
        return self.posteri


class NoneFilter:
    """utils.py:26-37"""

    def __init__(self):
        self.posteri = None

    def reset(self, mask=None):
        if mask is None:
            self.posteri = None

    def estimate(self, measurement=None):
        if measurement is not None:
            self.posteri = measurement
        return self.posteri


class ProcessNoise:
    """Ornstein-Uhlenbeck exploration noise (utils.py:69-86): theta 1, sigma sqrt(2), ten Euler
    sub-steps of dt 0.01 per call.  `normal(shape)` supplies the N(0,1) draws (tests feed numpy's
    legacy stream to reproduce the reference's numbers)."""

    def __init__(self, dim, batch=1, device="cpu", dtype=torch.float64, normal=None, generator=None):
        self.x = torch.zeros(batch, dim, device=device, dtype=dtype)
        self.t, self.n = 0.1, 10
        self.dt = self.t / self.n
        self.theta, self.sigma = 1.0, math.sqrt(2.0)
        self._normal = normal
        self._gen = generator

    def _draw(self):
        if self._normal is not None:
            return torch.as_tensor(self._normal(tuple(self.x.shape)), dtype=self.x.dtype, device=self.x.device)
        return torch.randn(self.x.shape, device=self.x.device, dtype=self.x.dtype, generator=self._gen)

    def next(self):
        for _ in range(self.n):
            self.x = self.x - self.theta * self.x * self.dt + self.sigma * self._draw() * math.sqrt(self.dt)
        return self.x


class ReplayBuffer:
    """FIFO ring (utils.py:40-66) held on the device; `store` takes one transition or a batch."""

    def __init__(self, obs_dim, act_dim, size, device="cpu"):
        f = dict(dtype=torch.float32, device=device)
        self.obs1_buf = torch.zeros(size, obs_dim, **f)
        self.obs2_buf = torch.zeros(size, obs_dim, **f)
        self.acts_buf = torch.zeros(size, act_dim, **f)
        self.rews_buf = torch.zeros(size, 1, **f)
        self.done_buf = torch.zeros(size, 1, **f)
        self.ptr, self.size, self.max_size = 0, 0, size
        self.device = device

    def store(self, obs, act, rew, next_obs, done):
        obs = torch.as_tensor(obs, dtype=torch.float32, device=self.device).reshape(-1, self.obs1_buf.shape[1])
        B = obs.shape[0]
        act = torch.as_tensor(act, dtype=torch.float32, device=self.device).reshape(B, -1)
        next_obs = torch.as_tensor(next_obs, dtype=torch.float32, device=self.device).reshape(B, -1)
        rew = torch.as_tensor(rew, dtype=torch.float32, device=self.device).reshape(-1).expand(B)
        done = torch.as_tensor(done, dtype=torch.float32, device=self.device).reshape(-1).expand(B)
        if B > self.max_size:  # only the newest max_size transitions survive, as with one-by-one stores
            raise ValueError("batch larger than the buffer")
        idx = (self.ptr + torch.arange(B, device=self.device)) % self.max_size
        self.obs1_buf[idx] = obs
        self.obs2_buf[idx] = next_obs
        self.acts_buf[idx] = act
        self.rews_buf[idx, 0] = rew
        self.done_buf[idx, 0] = done
        self.ptr = (self.ptr + B) % self.max_size
        self.size = min(self.size + B, self.max_size)

    def sample_batch(self, batch_size=32, generator=None):
        idxs = torch.randint(0, self.size, (batch_size,), device=self.device, generator=generator)
        return dict(obs1=self.obs1_buf[idxs], obs2=self.obs2_buf[idxs], acts=self.acts_buf[idxs],
                    rews=self.rews_buf[idxs], done=self.done_buf[idxs])

    def state_dict(self):
        n = self.size
        return dict(obs1=self.obs1_buf[:n].cpu(), obs2=self.obs2_buf[:n].cpu(), acts=self.acts_buf[:n].cpu(),
                    rews=self.rews_buf[:n].cpu(), done=self.done_buf[:n].cpu(), ptr=self.ptr, size=self.size,
                    max_size=self.max_size)

    def check_state_dict(self, sd):
        """raises if the checkpointed buffer cannot be restored into this one (nothing is modified)"""
        n = int(sd["size"])
        if n > self.max_size or n < 0:
            raise ValueError("replay buffer checkpoint holds %d transitions, this buffer %d at most" % (n, self.max_size))
        for key, buf in (("obs1", self.obs1_buf), ("obs2", self.obs2_buf), ("acts", self.acts_buf), ("rews", self.rews_buf),
                         ("done", self.done_buf)):
            if tuple(sd[key].shape) != (n,) + tuple(buf.shape[1:]):
                raise ValueError("replay buffer checkpoint: %s has shape %s, expected %s"
                                 % (key, tuple(sd[key].shape), (n,) + tuple(buf.shape[1:])))

    def load_state_dict(self, sd):
        """The reference restores by replacing the buffer object wholesale (server.py:192-198), so any pickled buffer
        restores there; here any checkpoint whose CONTENT fits is accepted, whatever capacity it was saved from: the
        saved transitions are the oldest-to-newest prefix and the write pointer follows them."""
        self.check_state_dict(sd)
        n = int(sd["size"])
        same = int(sd.get("max_size", self.max_size)) == self.max_size
        if same:
            order = torch.arange(n)
        else:  # a full buffer of another capacity is a ring: unroll it so that index 0 is the oldest transition
            old_max, ptr = int(sd.get("max_size", n)), int(sd["ptr"])
            order = (torch.arange(n) + (ptr if n == old_max else 0)) % max(n, 1)
        for name, key in (("obs1_buf", "obs1"), ("obs2_buf", "obs2"), ("acts_buf", "acts"), ("rews_buf", "rews"),
                          ("done_buf", "done")):
            getattr(self, name)[:n] = sd[key][order].to(self.device)
        self.size = n
        self.ptr = int(sd["ptr"]) % self.max_size if same else n % self.max_size


def _mlp(sizes, last):
    layers = []
    for i in range(len(sizes) - 1):
        lin = nn.Linear(sizes[i], sizes[i + 1])
        nn.init.xavier_uniform_(lin.weight)  # tf.layers.dense: glorot_uniform kernel, zero bias
        nn.init.zeros_(lin.bias)
        layers.append(lin)
        if i < len(sizes) - 2:
            layers.append(nn.ReLU())
    if last is not None:
        layers.append(last)
    return nn.Sequential(*layers)


class Agent:
    """agent_TD3.py:10-184.  `get_action`, `train_iter`, `record`, `reset_agent`, `reset_episode`
    keep the reference's names and meaning; states/actions may carry any batch size."""

    # Transitions per env and episode of the reference's case (clientCFD.pde: 16-step windows from t = 1 to t = 50 at
    # dt = 0.0075: 408 requests, 407 transitions): what a rank's replay ring has to hold per env to cover one whole
    # episode -- the DEFAULT; a trainer on another time stepping passes steps_per_episode (train.py: the request count of
    # its episode_plan).
    STEPS_PER_EPISODE = 410

    def __init__(self, state_dim, action_dim, explore_noise="Gaussian", device="cpu", seed=None, num_envs=1, world=1,
                 replay_capacity=None, replay_episodes=3, steps_per_episode=None):
        self.world = world  # > 1: sharded learner (gradients averaged over the ranks, see _sync_grads)
        self.num_envs = num_envs
        self.n_grad_allreduce = 0
        self.lr, self.gamma, self.tau = 1e-4, 0.99, 0.005
        self.bs, self.bfs, self.d = 512, 1000000, 2
        self.explore_noise = explore_noise
        self.explore_noise_size = 0.1
        self.criticreg_noise_size, self.criticreg_noise_clip = 0.2, 0.5
        self.state_dim, self.action_dim = state_dim, action_dim
        self.device = torch.device(device)
        self.gen = torch.Generator(device=self.device)
        if seed is not None:
            self.gen.manual_seed(seed)
            torch.manual_seed(seed)
        self.process_noise_generator = ProcessNoise(action_dim, batch=num_envs, device=self.device, dtype=torch.float32,
                                                    generator=self.gen)
        if world > 1 and self.bs % world:
            raise ValueError("sharded learner: the batch size %d must be a multiple of the world size %d" % (self.bs, world))
        self._build()
        if world > 1:
            self._sync_from_rank0(seed)
        # Replay capacity PER RANK (a rank stores the transitions of its own envs only, time-major): the reference's
        # 1 M (agent_TD3.py:27) spans ~2400 single-env episodes; a batched rank writes num_envs transitions per agent
        # step, so the ring must hold whole episodes of ALL its envs or training never sees the early part of an
        # episode (1024 envs x 410 steps = 420 k per episode: a 1 M / world share would keep the last ~120 agent steps
        # only).  max(1 M, replay_episodes x num_envs x STEPS_PER_EPISODE) transitions of 32 bytes each.
        # (job-wide that is world x this -- against the reference's single 1 M FIFO: a sampling deviation of the batched
        # trainer, recorded as `replay_capacity` in train.py's episode log)
        local_envs = max(1, int(num_envs) // max(int(world), 1))  # (num_envs counts the envs of ALL ranks)
        self.steps_per_episode = int(steps_per_episode) if steps_per_episode else self.STEPS_PER_EPISODE
        self.replay_capacity = int(replay_capacity) if replay_capacity else max(
            self.bfs, int(replay_episodes) * local_envs * self.steps_per_episode)
        self.replay_buffer = ReplayBuffer(state_dim, action_dim, self.replay_capacity, device=self.device)
        self.step_count = self.total_step_count = self.episode_count = self.train_count = 0

    def _nets(self):
        return (self.main_actor, self.target_actor, self.main_critic_1, self.main_critic_2, self.target_critic_1,
                self.target_critic_2)

    def _sync_from_rank0(self, seed):
        """sharded learner: every rank must start from the same weights and the same generator state (the ranks average
        gradients onto THEIR weights and draw the same exploration / target noise).  With a process group up, rank 0's
        are broadcast -- seeded or not; without one, only an explicit seed can guarantee it."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            if seed is None:
                raise ValueError("Agent(world > 1) needs a seed or an initialised torch.distributed process group: "
                                 "ranks that start from different weights never converge to one model")
            return
        for net in self._nets():
            for p in net.parameters():
                dist.broadcast(p.data, 0)
        st = self.gen.get_state().to(self.device)
        dist.broadcast(st, 0)
        self.gen.set_state(st.cpu())

    def _build(self):
        s, a, dev = self.state_dim, self.action_dim, self.device
        self.main_actor = _mlp([s, 256, 256, a], nn.Tanh()).to(dev)
        self.target_actor = _mlp([s, 256, 256, a], nn.Tanh()).to(dev)
        self.main_critic_1 = _mlp([s + a, 256, 256, 1], None).to(dev)
        self.main_critic_2 = _mlp([s + a, 256, 256, 1], None).to(dev)
        self.target_critic_1 = _mlp([s + a, 256, 256, 1], None).to(dev)
        self.target_critic_2 = _mlp([s + a, 256, 256, 1], None).to(dev)
        self._pairs = [(self.target_critic_1, self.main_critic_1), (self.target_critic_2, self.main_critic_2),
                       (self.target_actor, self.main_actor)]
        for t, m in self._pairs:  # T_init
            t.load_state_dict(m.state_dict())
            for p in t.parameters():
                p.requires_grad_(False)
        critic_params = list(self.main_critic_1.parameters()) + list(self.main_critic_2.parameters())
        self._critic_params = critic_params
        self.Q_opt = torch.optim.Adam(critic_params, lr=self.lr, eps=1e-8)
        self.P_opt = torch.optim.Adam(self.main_actor.parameters(), lr=self.lr, eps=1e-8)

    # -- acting --
    @torch.no_grad()
    def get_action(self, state_data, stochastic=True):
        state = torch.as_tensor(state_data, dtype=torch.float32, device=self.device).reshape(-1, self.state_dim)
        action = self.main_actor(state)
        if stochastic:
            if self.explore_noise == "Gaussian":
                noise = self.explore_noise_size * torch.randn(action.shape, device=self.device, generator=self.gen)
            elif self.explore_noise == "Process":
                if self.process_noise_generator.x.shape[0] != action.shape[0]:
                    self.process_noise_generator = ProcessNoise(self.action_dim, batch=action.shape[0], device=self.device,
                                                                dtype=torch.float32, generator=self.gen)
                noise = self.explore_noise_size * self.process_noise_generator.next()
            else:
                raise NotImplementedError
            action = torch.clamp(action + noise, -1, 1)
        return action

    # -- learning --
    def losses(self, batch):
        """(Q_loss, target y) of agent_TD3.py:36-48 for a batch dict."""
        with torch.no_grad():
            nxt = self.target_actor(batch["obs2"])
            noise = torch.clamp(self.criticreg_noise_size * torch.randn(nxt.shape, device=self.device, generator=self.gen),
                                -self.criticreg_noise_clip, self.criticreg_noise_clip)
            nxt = torch.clamp(nxt + noise, -1, 1)
            sa2 = torch.cat([batch["obs2"], nxt], 1)
            q_min = torch.minimum(self.target_critic_1(sa2), self.target_critic_2(sa2))
            y = batch["rews"] + self.gamma * (1 - batch["done"]) * q_min
        sa1 = torch.cat([batch["obs1"], batch["acts"]], 1)
        q_loss = ((self.main_critic_1(sa1) - y) ** 2).mean() + ((self.main_critic_2(sa1) - y) ** 2).mean()
        return q_loss, y

    def _sync_grads(self, params):
        """sharded learner: every rank holds the transitions of its own envs and a 1/world share of the batch; the
        gradients are averaged over the ranks (one all-reduce of the flattened gradients), so every rank applies the
        same update to the same weights"""
        if self.world <= 1:
            return
        import torch.distributed as dist
        params = [p for p in params if p.grad is not None]
        flat = torch.cat([p.grad.reshape(-1) for p in params])
        dist.all_reduce(flat)
        flat /= self.world
        o = 0
        for p in params:
            n = p.grad.numel()
            p.grad.copy_(flat[o:o + n].view_as(p.grad))
            o += n
        self.n_grad_allreduce += 1

    def train_iter(self):
        bs = self.bs // max(self.world, 1)  # this rank's share of the batch (bs is a multiple of world: __init__)
        if bs > self.replay_buffer.size:
            return
        batch = self.replay_buffer.sample_batch(bs, generator=self.gen)
        q_loss, _ = self.losses(batch)
        self.Q_opt.zero_grad(set_to_none=True)
        q_loss.backward()
        self._sync_grads(self._critic_params)
        self.Q_opt.step()
        if self.total_step_count % self.d == 0:
            p_loss = -self.main_critic_1(torch.cat([batch["obs1"], self.main_actor(batch["obs1"])], 1)).mean()
            self.P_opt.zero_grad(set_to_none=True)
            p_loss.backward()
            self._sync_grads(list(self.main_actor.parameters()))
            self.P_opt.step()
            with torch.no_grad():  # T_op
                for t, m in self._pairs:
                    for pt, pm in zip(t.parameters(), m.parameters()):
                        pt.mul_(1 - self.tau).add_(pm, alpha=self.tau)
        self.train_count += 1
        self.total_step_count += 1

    def record(self, this_state, this_action, this_reward, this_done, next_state):
        self.replay_buffer.store(obs=this_state, act=this_action, rew=this_reward, next_obs=next_state, done=this_done)
        self.step_count += 1

    def reset_agent(self):
        self.replay_buffer = ReplayBuffer(self.state_dim, self.action_dim, self.replay_capacity, device=self.device)
        self.step_count = self.total_step_count = self.train_count = self.episode_count = 0
        self._build()
        if self.world > 1:
            self._sync_from_rank0(0)  # a fresh model on every rank: rank 0's where a process group is up

    def reset_episode(self):
        self.step_count = 0
        self.train_count = 0
        self.episode_count += 1

    # -- checkpoint (server.py:181-198 saves a TF checkpoint + the pickled buffer) --
    def state_dict(self):
        nets = {n: getattr(self, n).state_dict() for n in ("main_actor", "target_actor", "main_critic_1", "main_critic_2",
                                                           "target_critic_1", "target_critic_2")}
        return dict(nets=nets, Q_opt=self.Q_opt.state_dict(), P_opt=self.P_opt.state_dict(),
                    replay=self.replay_buffer.state_dict(), episode_count=self.episode_count,
                    total_step_count=self.total_step_count)

    def load_state_dict(self, sd):
        self.replay_buffer.check_state_dict(sd["replay"])  # validate everything before the first mutation
        for n in sd["nets"]:
            if not hasattr(self, n):
                raise ValueError("checkpoint has an unknown network %r" % n)
        for n, s in sd["nets"].items():
            getattr(self, n).load_state_dict(s)
        self.Q_opt.load_state_dict(sd["Q_opt"])
        self.P_opt.load_state_dict(sd["P_opt"])
        self.replay_buffer.load_state_dict(sd["replay"])
        self.episode_count = sd["episode_count"]
        # total_step_count is NOT restored: in the reference it is a plain Python attribute outside the TF
        # checkpoint and the pickled buffer (server.py:192-198, agent_TD3.py:167-173), so a restored agent
        # restarts its policy-delay phase from 0; the saved value is kept in the file for information only


def assemble_transitions(state_record, action_record, action_dim):
    """server.py:157-165: consecutive recorded (filtered) states and actions of ONE episode become
    transitions (s_i, a_i[:action_dim], r_i, s_{i+1}, done = 0), i = 0 .. len-2.
    state_record [T, ..., 2], action_record [T, ..., A] (any batch dims in the middle)."""
    s = torch.as_tensor(state_record)
    a = torch.as_tensor(action_record)
    s1, s2, act = s[:-1], s[1:], a[:-1]
    r = reward_func(s2, act)
    return s1, act[..., :action_dim], r, s2, torch.zeros_like(r)


class BatchedLearner:
    """The reference's episode protocol (clientCFD.pde:35-85 + server.py) for E envs stepping together:
    act on the filtered obs of every env each agent step, keep the episode's records on the device,
    and at the end of the episode turn them into transitions and run `train_steps` TD3 iterations."""

    def __init__(self, num_envs, device="cpu", filter="None", explore_noise="Gaussian", one_action=False, seed=None,
                 shard=None, steps_per_episode=None):
        """num_envs = ALL envs this learner acts for (every rank evaluates the same actor on the gathered obs).
        shard = (rank, world): the learner is sharded -- this rank stores the transitions of ITS envs only (the block
        [rank*E, (rank+1)*E) of the env axis) and trains on a 1/world share of every batch with averaged gradients."""
        self.E = num_envs
        self.rank, self.world = shard if shard is not None else (0, 1)
        self.one_action = one_action
        self.action_dim = 1 if one_action else 2
        self.agent = Agent(2, self.action_dim, explore_noise=explore_noise, device=device, seed=seed, num_envs=num_envs,
                           world=self.world, steps_per_episode=steps_per_episode)
        self.filter = KalmanFilter() if filter == "Kalman" else NoneFilter()
        self.state_record, self.action_record = [], []

    def start_episode(self):
        self.state_record, self.action_record = [], []
        self.agent.reset_episode()
        self.filter.reset()

    def request_action(self, obs, stochastic=True):
        """obs [E,2] = (Cl, Cd) -> action [E,2] in [-1,1]"""
        filtered = self.filter.estimate(obs)
        action = self.agent.get_action(filtered, stochastic=stochastic)
        if self.one_action:
            action = torch.cat([action, -action], 1)
        self.state_record.append(filtered.clone())
        self.action_record.append(action.clone())
        return action

    def train(self, steps):
        if len(self.state_record) >= 2:
            s1, act, r, s2, done = assemble_transitions(torch.stack(self.state_record), torch.stack(self.action_record),
                                                        self.action_dim)
            if self.world > 1:  # this rank keeps its own envs' transitions
                per = self.E // self.world
                sl = slice(self.rank * per, (self.rank + 1) * per)
                s1, act, r, s2, done = s1[:, sl], act[:, sl], r[:, sl], s2[:, sl], done[:, sl]
            # time-major records -> one store per agent step keeps the reference's FIFO order per env
            for t in range(s1.shape[0]):
                self.agent.replay_buffer.store(s1[t], act[t], r[t], s2[t], done[t])
        for _ in range(steps):
            self.agent.train_iter()
