"""Per-phase wall time of one env's sim-step while E envs run (diagnostic, GPU box)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlfluidcontrol_amd import build, env  # noqa: E402

build.build()
E = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
res = int(sys.argv[2]) if len(sys.argv) > 2 else 16
nst = int(sys.argv[3]) if len(sys.argv) > 3 else 8
impl = int(sys.argv[4]) if len(sys.argv) > 4 else 2
g = env.AFCEnv(num_envs=E, resolution=res, reynolds=100, impl=impl)
xi = np.zeros((E, 2), np.float32)
for _ in range(4):
    g.update2(xi)
g.profile_phases(True)
import time
t0 = time.time()
rng = np.random.default_rng(0)
xi = (5 * rng.uniform(-1, 1, (E, 2))).astype(np.float32)
for _ in range(nst):
    g.update2(xi)
dt = time.time() - t0
ph = g.profile_phases(False)
tot = sum(ph.values())
print("E=%d res=%d: %.3f ms per sim-step (host), env0 phases sum %.3f ms" % (E, res, 1e3 * dt / nst, tot / nst / 1e3))
for k, v in ph.items():
    print("  %-12s %9.1f us/step  %5.1f%%" % (k, v / nst, 100 * v / max(tot, 1)))
