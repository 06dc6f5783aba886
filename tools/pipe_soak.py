"""Race detector for the pipelined up-leg under full load: 1024 envs, random actions, N agent steps, run (a) twice through the
shipped kernels (any timing-dependent hazard shows as a difference between two runs of the same code on the same inputs) and
(b) in parity mode through the shipped kernels and through the barrier form (dbg bit 8, diagnostic instantiation): every
observation of every env at every agent step and the final fields of 64 envs must be equal bit for bit.
(c) the same for the up-leg that marches down the rows (upleg = 3, the instantiation small batches get) at one workgroup per CU
(256 envs, where the host picks it) and forced at four per CU (1024 envs, where its helpers are starved the most).
    python tools/pipe_soak.py [agent steps]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlfluidcontrol_amd import build, env  # noqa: E402

build.build()
E = 1024
N = int(sys.argv[1]) if len(sys.argv) > 1 else 24
z = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "wake_res16_re100.npz"))


def run(dbg, exact, upleg=1, E=E):
    g = env.AFCEnv(num_envs=E, resolution=16, reynolds=100, dbg=dbg, exact_sums=exact, upleg=upleg)
    assert g.upleg() == upleg
    g.set_reset_state(z["u"], z["v"], z["p"])
    g.reset()
    rng = np.random.default_rng(7)
    rec = []
    for _ in range(N):
        a = rng.uniform(-1, 1, (E, 2)).astype(np.float32)
        obs, rew, done, ns = g.agent_step(a)
        rec.append((obs.copy(), rew.copy(), ns.copy()))
    st = [g.get_state(e) for e in range(0, E, 16)]
    h = g.mg_histogram()
    g.close()
    return rec, st, h


def same(a, b, what):
    (ra, sa, ha), (rb, sb, hb) = a, b
    assert np.array_equal(ha, hb), (what, ha, hb)
    for k, ((o1, w1, n1), (o2, w2, n2)) in enumerate(zip(ra, rb)):
        assert np.array_equal(n1, n2) and np.array_equal(o1.view(np.uint32), o2.view(np.uint32)) and \
            np.array_equal(w1.view(np.uint32), w2.view(np.uint32)), (what, "agent step", k)
    for e, (x, y) in enumerate(zip(sa, sb)):
        for f1, f2 in zip(x, y):
            assert np.array_equal(f1.view(np.uint32), f2.view(np.uint32)), (what, "env", 16 * e)
    print("%-70s identical (%d agent steps, MG iterations %s)" % (what, N, {i + 1: int(v) for i, v in enumerate(ha) if v}))


a1 = run(0, 0)
a2 = run(0, 0)
same(a1, a2, "shipped kernels, default mode, two runs")
p1 = run(0, 1)
p2 = run(256, 1)
same(p1, p2, "parity mode: pipelined up-leg against the barrier form (dbg bit 8)")
p3 = run(0, 1)
same(p1, p3, "parity mode, two runs")
for envs in (256, 1024):
    m1 = run(0, 0, 3, envs)
    m2 = run(0, 0, 3, envs)
    same(m1, m2, "%d envs, rows march (upleg 3), default mode, two runs" % envs)
    same(m1, run(0, 0, 1, envs), "%d envs, default mode: rows march against the strip pipeline" % envs)
    q1 = run(0, 1, 3, envs)
    same(q1, run(256, 1, 1, envs), "%d envs, parity mode: rows march against the barrier form (dbg bit 8)" % envs)
