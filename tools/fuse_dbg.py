"""development: the fused corrector advection (AFC_FUSE_TAIL=1) against the two-pass form, field by field after a few
sim-steps: where do they differ?   python tools/fuse_dbg.py [res] [envs] [steps]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlfluidcontrol_amd import build, env  # noqa: E402

build.build()
res = int(sys.argv[1]) if len(sys.argv) > 1 else 16
E = int(sys.argv[2]) if len(sys.argv) > 2 else 1
nst = int(sys.argv[3]) if len(sys.argv) > 3 else 1
xi = np.tile(np.array([[1.5, -2.0]], np.float32), (E, 1))


def run(fuse):
    g = env.AFCEnv(num_envs=E, resolution=res, reynolds=100, fuse_tail=(1 if fuse else -1))
    print("fused:", g.fused_tail())
    g.reset()
    for _ in range(nst):
        g.update2(xi)
    for k in range(3):  # launches of several sim-steps: the Heun tail of all but the last one is left to the next step
        o, r, d, ns = g.agent_step(xi / 5)
        print("  agent step", k, "sim-steps", ns[0], "obs", o[0], "rew", r[0], "all envs alike:", bool((o == o[0]).all()))
    out = [np.array(g.get_field(k, e)) for e in sorted({0, E - 1, E // 2}) for k in (env.FIELD_U, env.FIELD_V, env.FIELD_P)]
    g.close()
    return out


a, b = run(0), run(1)  # two-pass | fused
for k, (x, y) in enumerate(zip(a, b)):
    d = np.argwhere(x != y)
    print("field", k, x.shape, "differing cells:", len(d), "rows:", sorted(set(d[:, 0]))[:12], "cols:", sorted(set(d[:, 1]))[:12])
    if len(d):
        i, j = d[0]
        print("   first", i, j, x[i, j], y[i, j])
