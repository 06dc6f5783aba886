"""Contention inside a CU or for the chip?  The same 512 envs two per CU on all 256 CUs, and four per CU on half the CUs
(the other workgroups of a 1024-env launch leave at once: gate value 14), next to 1024 envs four per CU on all CUs.
    python tools/placement_test.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlfluidcontrol_amd import build, env  # noqa: E402

build.build()
z = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "wake_res16_re100.npz"))


def run(E, dbg, label):
    g = env.AFCEnv(num_envs=E, resolution=16, reynolds=100, dbg=dbg)
    g.set_reset_state(z["u"], z["v"], z["p"])
    g.reset()
    rng = np.random.default_rng(0)
    a = rng.uniform(-1, 1, (E, 2)).astype(np.float32)
    for _ in range(3):
        g.agent_step(a)
    best = 1e9
    for _ in range(6):
        t0 = time.perf_counter()
        g.agent_step(a)
        best = min(best, time.perf_counter() - t0)
    print("%-58s %7.2f ms per agent step" % (label, 1e3 * best), flush=True)
    g.close()


run(256, 1 << 30, "256 envs, one per CU")
run(512, 1 << 30, "512 envs, two per CU on all CUs")
run(1024, 14 << 16, "512 working envs, four per CU on half the CUs")
run(1024, 1 << 30, "1024 envs, four per CU on all CUs")
