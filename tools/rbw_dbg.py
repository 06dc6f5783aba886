"""development: the perf-mode up-leg as register-resident row pipelines (mg_up_rbw) against the barrier form, level by level
(AFC_RBW = mask of levels), one V-cycle of afc_op_mg_solve on a random right-hand side"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlfluidcontrol_amd import build, env
build.build()
res = int(sys.argv[1]) if len(sys.argv) > 1 else 8
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 1
def run(mask):
    os.environ["AFC_RBW"] = str(mask)
    g = env.AFCEnv(num_envs=1, resolution=res, reynolds=100, gs_mode=1, mg_max_iters=iters)
    rng = np.random.default_rng(3)
    b = np.zeros((g.n, g.m), np.float32)
    b[1:-1, 1:-1] = 1e-2 * rng.standard_normal((g.n - 2, g.m - 2)).astype(np.float32)
    b[1:-1, 1:-1] -= b[1:-1, 1:-1].mean()
    x, it, l = g.op_mg_solve(np.zeros_like(b), b)
    g.close()
    return x, it, l
x0, it0, l0 = run(0)
print("barrier form: iters", it0, "linf", l0)
nlev = 7
for mask in [1 << k for k in range(nlev)] + [(1 << nlev) - 1]:
    x, it, l = run(mask)
    d = np.abs(x - x0)
    print("mask %3d: iters %d linf %g  max|dx| %g  nan %d  same %s" % (mask, it, l, np.nanmax(d), int(np.isnan(x).sum()), np.array_equal(x, x0)))
    if not np.array_equal(x, x0) and mask == 1:
        rows = d.max(axis=1)
        print("   per-row max|dx| (rows 0..):", " ".join("%.1e" % v for v in rows[:40]))
        print("   ...", " ".join("%.1e" % v for v in rows[40:80]))
        cols = d.max(axis=0)
        print("   per-col max|dx|:", " ".join("%.1e" % v for v in cols[:34]))
