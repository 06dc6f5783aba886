"""Soak run: 1024 envs through a whole episode (t = 0 .. 50, ~417 agent steps from the developed wake)
with random actions: no env may diverge, every env must finish together, reset must restart cleanly."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlfluidcontrol_amd import build, env  # noqa: E402

build.build()
E = 1024
g = env.AFCEnv(num_envs=E, resolution=16, reynolds=100)
z = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "wake_res16_re100.npz"))
g.set_reset_state(z["u"], z["v"], z["p"])
g.reset()
rng = np.random.default_rng(0)
t0 = time.time()
n_req, sim_steps = 0, 0
cd = []
while True:
    a = rng.uniform(-1, 1, (E, 2)).astype(np.float32)
    obs, rew, done, ns = g.agent_step(a)
    sim_steps += int(ns[0])
    assert (ns == ns[0]).all() and (done == done[0]).all()
    if done[0]:
        break
    assert np.isfinite(obs).all() and np.isfinite(rew).all()
    cd.append(obs[:, 1].mean())
    n_req += 1
dt = time.time() - t0
it, rs = g.solver_stats()
print("episode: %d requests, %d sim-steps per env, %.1f s -> %.0f env-sim-steps/s incl. host I/O" % (n_req, sim_steps, dt, E * sim_steps / dt))
print("diverged envs: %d; t = %.4f; mean Cd first/last window: %.4f / %.4f; max MG iterations seen at the end: %d" % (
    int(g.diverged().sum()), float(g.t[0]), cd[0], cd[-1], int(it.max())))
u0 = g.get_state(0)[0]
u1 = g.get_state(E - 1)[0]
print("envs decorrelated by their own actions: max |u_0 - u_last| = %.3f" % float(np.abs(u0 - u1).max()))
g.reset()
obs, rew, done, ns = g.agent_step(np.zeros((E, 2), np.float32))
assert not done.any() and (ns == ns[0]).all() and np.isfinite(obs).all()
print("after reset: first window has %d sim-steps, Cd %.4f" % (int(ns[0]), float(obs[0, 1])))
