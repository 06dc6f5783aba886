"""Where in an agent step do second V-cycles occur?  1024 envs from the developed wake, a new random action every 16
sim-steps, multigrid iterations of both pressure solves recorded per sim-step (GPU box):  python tools/iters_by_step.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlfluidcontrol_amd import build, env
build.build()
E = 1024
g = env.AFCEnv(num_envs=E, resolution=16, reynolds=100)
z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "wake_res16_re100.npz"))
g.set_reset_state(z["u"], z["v"], z["p"]); g.reset()
rng = np.random.default_rng(0)
hist = np.zeros((16, 2, 4), np.int64)
for a in range(12):
    xi = (5 * rng.uniform(-1, 1, (E, 2))).astype(np.float32)
    for k in range(16):
        g.update2(xi)
        it, _ = g.solver_stats()
        if a >= 2:
            for h in range(2):
                for v in range(1, 5):
                    hist[k, h, v - 1] += int((it[:, h] == v).sum())
print("solves by (sim-step within the agent step, half-step): count of 1 / 2 / 3 / 4 V-cycles")
for k in range(16):
    print(k, hist[k, 0], hist[k, 1])
