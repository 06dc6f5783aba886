"""Times the step kernel (E envs, stop code `stop`) under several StepParams.dbg experiment switches in ONE process:
    python tools/ab_dbg.py <stop> <dbg1> <dbg2> ... [E=1024]
prints the min-of-8 time of one sim-step launch per switch value (diagnostic builds only; results are not physical)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlfluidcontrol_amd import build, env  # noqa: E402

build.build()
stop = int(sys.argv[1])
dbgs = [int(x) for x in sys.argv[2:] if "=" not in x]
E = 1024
for x in sys.argv[2:]:
    if x.startswith("E="):
        E = int(x[2:])
z = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "wake_res16_re100.npz"))
rng = np.random.default_rng(0)
xi = (5 * rng.uniform(-1, 1, (E, 2))).astype(np.float32)
for d in dbgs:
    g = env.AFCEnv(num_envs=E, resolution=16, reynolds=100, dbg=(stop << 12) | d)
    g.set_reset_state(z["u"], z["v"], z["p"])
    g.reset()
    for _ in range(3):
        g.update2(xi)
    best = 1e9
    for _ in range(8):
        t0 = time.perf_counter()
        g.update2(xi)
        best = min(best, time.perf_counter() - t0)
    print("stop %d dbg %4d: %8.1f us" % (stop, d, best * 1e6), flush=True)
    g.close()
