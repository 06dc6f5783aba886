"""Lock-step cost of each phase of one sim-step with E envs resident: the step kernel is launched with a
`stop after phase` code (StepParams.dbg bits 12-15) and timed; differences of consecutive codes are the
time the WHOLE batch needs for that phase when every workgroup is in it at once (the throughput view;
tools/phase_profile.py gives the one-env latency view).
    python tools/phase_times.py [E] [res]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlfluidcontrol_amd import build, env  # noqa: E402

STOPS = [1, 2, 3, 4, 5, 6, 9, 10, 11, 12, 13, 14, 0]
NAMES = ["advdif", "forcing+setbc", "head(down0)", "coarse levels", "up0 (L0 fill/march/exit)", "project tail"]


def main():
    build.build()
    E = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    res = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    # the committed start state of the grid (developed wake / the reference's init.bdim) and its Reynolds number
    gold = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")
    name, Re = {16: ("wake_res16_re100.npz", 100), 32: ("wake_res32_re400.npz", 400), 24: ("init_bdim_res24.npz", 500)}.get(res, (None, 100))
    z = np.load(os.path.join(gold, name)) if name and os.path.exists(os.path.join(gold, name)) else None
    rng = np.random.default_rng(0)
    xi = (5 * rng.uniform(-1, 1, (E, 2))).astype(np.float32)
    ts = []
    for st in STOPS:
        g = env.AFCEnv(num_envs=E, resolution=res, reynolds=Re, dbg=st << 12, upleg=int(os.environ.get("UPLEG", "0")))
        if z is not None:
            g.set_reset_state(z["u"], z["v"], z["p"])
            g.reset()
        for _ in range(3):
            g.update2(xi)
        best = 1e9
        for _ in range(6):
            t0 = time.perf_counter()
            g.update2(xi)
            best = min(best, time.perf_counter() - t0)
        ts.append(best * 1e6)
        g.close()
    prev = 0.0
    print("E=%d res=%d: lock-step phase times (us per sim-step, min of 6; first row includes launch + I/O overhead)" % (E, res))
    for i, st in enumerate(STOPS):
        if st == 0:
            print("%-34s %9.1f   (whole step %9.1f)" % ("force + bookkeeping", ts[i] - prev, ts[i]))
            continue
        half, ph = divmod(st - 1, 8)
        print("%-34s %9.1f" % ("half %d %s" % (half, NAMES[ph]), ts[i] - prev))
        prev = ts[i]


if __name__ == "__main__":
    main()
