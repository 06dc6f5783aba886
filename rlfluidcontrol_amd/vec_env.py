"""Env-sharded vector environment: one process per GPU, E envs per rank, no data-path
collective except ONE all-gather of (obs, reward, done) per agent step.

Envs are independent simulations, so sharding is a partition of the env index: global env
g = rank*E + e.  `backend` is any object with `agent_step_device`-like semantics; the default is
the HIP `AFCEnv`.  The gather packs obs[2] | reward | done into one [E,4] f32 buffer so that a
single RCCL (or gloo, in the CPU tests) call moves 16 B/env.
"""
import torch
import torch.distributed as dist


class ShardedVecEnv:
    def __init__(self, backend, num_envs_per_rank, device=None):
        self.backend = backend
        self.E = num_envs_per_rank
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.device = device if device is not None else torch.device("cpu")
        self.packed = torch.zeros(self.E, 4, dtype=torch.float32, device=self.device)
        self.gathered = torch.zeros(self.world * self.E, 4, dtype=torch.float32, device=self.device)
        self.n_collectives = 0

    def global_index(self, e):
        return self.rank * self.E + e

    def local_slice(self):
        return slice(self.rank * self.E, (self.rank + 1) * self.E)

    def step(self, global_actions):
        """global_actions [world*E, 2] (every rank holds the same tensor, e.g. computed by a
        replicated actor on the gathered obs).  Returns global (obs [W*E,2], reward [W*E], done [W*E])."""
        local = global_actions[self.local_slice()].contiguous()
        obs, rew, done = self.backend.agent_step_tensors(local)
        self.packed[:, 0:2] = obs
        self.packed[:, 2] = rew
        self.packed[:, 3] = done.to(torch.float32)
        if self.world > 1:
            dist.all_gather_into_tensor(self.gathered, self.packed)
            self.n_collectives += 1
        else:
            self.gathered.copy_(self.packed)
        g = self.gathered
        return g[:, 0:2], g[:, 2], g[:, 3] > 0.5

    def window_complete(self):
        """True if the last step ended with a completed averaging window (lock-step episodes: env 0 of this
        rank speaks for all; only consulted on the step that ends an episode)"""
        f = getattr(self.backend, "window_complete", None)
        return bool(f()) if f is not None else False

    def reset_done(self, done_global):
        self.backend.reset_mask(done_global[self.local_slice()].to(torch.uint8).contiguous())


class HipBackend:
    """Adapter of AFCEnv to the tensor interface ShardedVecEnv expects (device pointers, async)."""

    def __init__(self, env, device):
        self.env = env
        self.device = device
        E = env.num_envs
        self.obs = torch.zeros(E, 2, device=device)
        self.rew = torch.zeros(E, device=device)
        self.done = torch.zeros(E, dtype=torch.uint8, device=device)
        self.nst = torch.zeros(E, dtype=torch.int32, device=device)
        env.use_device_io(torch.cuda.current_stream(device).cuda_stream)

    def agent_step_tensors(self, actions):
        a = actions.to(self.device, torch.float32).contiguous()
        self.env.agent_step_device(a.data_ptr(), self.obs.data_ptr(), self.rew.data_ptr(), self.done.data_ptr(),
                                   self.nst.data_ptr())
        return self.obs, self.rew, self.done

    def reset_mask(self, mask_u8):
        self.env.reset(mask_u8.data_ptr())

    def window_complete(self):
        return bool(self.env.window_complete()[0])
