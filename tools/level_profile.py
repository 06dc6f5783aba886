"""fill / march / exit time of the up-leg of every multigrid level inside env 0's sim-step with E envs resident
(phase marks of the level with n rows: dbg bits 20+; the marks of level 0 are always on, so a level's figures are the
difference to the run without extra marks).   python tools/level_profile.py [E] [steps]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlfluidcontrol_amd import build, env  # noqa: E402

build.build()
E = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nst = int(sys.argv[2]) if len(sys.argv) > 2 else 8
z = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "wake_res16_re100.npz"))
rng = np.random.default_rng(0)
xi = (5 * rng.uniform(-1, 1, (E, 2))).astype(np.float32)


def run(nrows):
    g = env.AFCEnv(num_envs=E, resolution=16, reynolds=100, dbg=(nrows << 20) | (1 << 30 if nrows == 0 else 0), upleg=int(os.environ.get("UPLEG", "0")))
    g.set_reset_state(z["u"], z["v"], z["p"])
    g.reset()
    for _ in range(3):
        g.update2(xi)
    g.profile_phases(True)
    for _ in range(nst):
        g.update2(xi)
    ph = g.profile_phases(False)
    g.close()
    return {k: v / nst for k, v in ph.items()}


base = run(0)
print("E=%d: env 0, us per sim-step (both half-steps); a level whose up-leg runs as the software pipeline (mg_up_pipe) has ONE mark:" % E)
print("its whole up-leg is listed under `march`, fill and exit run beside it")
print("%-8s %8s %8s %8s %8s" % ("level", "fill", "march", "exit", "whole"))
print("%-8s %8.1f %8.1f %8.1f %8.1f" % ("L0", base["up0"], base["gs0"], base["inc0"], base["up0"] + base["gs0"] + base["inc0"]))
for l, n in enumerate([130, 66, 34, 18, 10, 6], start=1):
    p = run(n)
    f, m_, x = p["up0"] - base["up0"], p["gs0"] - base["gs0"], p["inc0"] - base["inc0"]
    print("%-8s %8.1f %8.1f %8.1f %8.1f   (up_L%d total %.1f, down_L%d %.1f)" % ("L%d" % l, f, m_, x, f + m_ + x, l, base.get("up_L%d" % l, 0), l, base.get("down_L%d" % l, 0)))
print({k: round(v, 1) for k, v in base.items()})
