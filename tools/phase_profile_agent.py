"""Per-phase time of env 0 inside whole agent-step launches (16 sim-steps per launch), optional start stagger."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlfluidcontrol_amd import build, env  # noqa: E402

build.build()
E = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
stag = int(sys.argv[2], 0) if len(sys.argv) > 2 else 0
nag = 3
g = env.AFCEnv(num_envs=E, resolution=16, reynolds=100, stagger_us=stag)
rng = np.random.default_rng(0)
act = (0.5 * rng.uniform(-1, 1, (E, 2))).astype(np.float32)
g.agent_step(act)
g.profile_phases(True)
t0 = time.time()
n = 0
for _ in range(nag):
    n += int(g.agent_step(act)[3][0])
dt = time.time() - t0
ph = g.profile_phases(False)
tot = sum(ph.values())
print("E=%d stagger=%#x: %.3f ms per sim-step (host), env0 phases sum %.3f ms over %d steps" % (E, stag, 1e3 * dt / n, tot / n / 1e3, n))
print("  " + "  ".join("%s %.0f" % (k, v / n) for k, v in ph.items() if v))
