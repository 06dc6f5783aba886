import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from rlfluidcontrol_amd import env
for E in (16, 64, 1024):
    for dbg in (0, 15 << 16):
        g = env.AFCEnv(num_envs=E, resolution=16, reynolds=100, dbg=dbg)
        xi = np.zeros((E, 2), np.float32)
        g.update2(xi)
        t0 = time.perf_counter()
        for _ in range(4):
            g.update2(xi)
        dt = (time.perf_counter() - t0) / 4
        print("E=%d dbg=%d: %.3f ms per sim-step" % (E, dbg, dt * 1e3), flush=True)
        g.close()
