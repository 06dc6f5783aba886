"""Long-horizon parity is STATISTICAL ("chaotic divergence"): over several vortex-shedding periods
the HIP path is compared with the CPU oracle through the quantities the agent sees (clientCFD.pde:39-47 averages the
forces) -- mean Cd, rms of the fluctuating Cl and the shedding period (Strouhal number) from the zero crossings of
Cl - mean(Cl) -- in the DEFAULT mode (tree sums: what bench.py and train.py run) and in the PERF mode (red-black
smoother), under zero action and under a fixed non-zero action, on the two grids the oracle can follow in seconds.

Stated bounds: |mean Cd| 0.5 %, rms Cl 2 %, shedding period 1 %.  Measured on MI355X: see the
numbers the test prints (pytest -s).
"""
import os

import numpy as np
import pytest

from tests.conftest import GOLDEN

pytestmark = pytest.mark.gpu

N_STEPS = 4500        # >= 4 shedding periods (a period is ~785-840 sim-steps at res 16, ~885-910 at res 8; Re 100)
SPINUP_RES8 = 9000    # res 8 has no committed developed wake: spun up on the GPU inside the test
TOL_CD, TOL_CL, TOL_ST = 5e-3, 2e-2, 1e-2


@pytest.fixture(scope="module")
def envmod():
    from rlfluidcontrol_amd import build, env
    build.build()
    return env


def _stats(f, res):
    """(mean Cd, rms of Cl - mean Cl, shedding period in sim-steps) of a force history f [N,2]; coefficients as the
    reference normalises them (Cd = Fx * 2 / D, clientCFD.pde:44-45)"""
    cd = f[:, 0].astype(np.float64) * 2.0 / res
    cl = f[:, 1].astype(np.float64) * 2.0 / res
    y = cl - cl.mean()
    up = np.nonzero((y[:-1] < 0) & (y[1:] >= 0))[0]
    t = up + (-y[up]) / (y[up + 1] - y[up])  # linear interpolation of the upward zero crossings
    period = (t[-1] - t[0]) / (len(t) - 1) if len(t) >= 3 else float("nan")
    return cd.mean(), float(np.sqrt(np.mean(y * y))), period, len(t)


def _developed_wake(envmod, res):
    if res == 16:
        z = np.load(os.path.join(GOLDEN, "wake_res16_re100.npz"))
        return z["u"], z["v"], z["p"]
    g = envmod.AFCEnv(num_envs=1, resolution=res, reynolds=100, end_time=1e9, init_time=-1.0)
    # an asymmetric kick, then free development: the symmetric start would take far longer to begin shedding
    g.update2(np.array([[5.0, 5.0]], np.float32))
    done = 0
    while done < SPINUP_RES8:
        _, _, _, ns = g.agent_step(np.zeros((1, 2), np.float32))
        done += int(ns[0])
    st = g.get_state(0)
    g.close()
    return st


def _gpu_history(envmod, res, state, xi, gs_mode):
    g = envmod.AFCEnv(num_envs=2, resolution=res, reynolds=100, gs_mode=gs_mode)
    g.set_reset_state(*state)
    g.reset()
    a = np.tile(np.array([xi], np.float32), (2, 1))
    f = np.empty((N_STEPS, 2), np.float32)
    for k in range(N_STEPS):
        f[k] = g.update2(a)[1]
    assert not g.diverged().any()
    g.close()
    return f


# fixed non-zero actions (xi_m = act_scale * a, AFCCylinder.pde:48-49): on the 128x64 grid the control cylinders are one
# cell across and a strong rotation stretches the shedding period beyond the window, so that grid gets a milder one
ACTIONS = {8: [(0.0, 0.0), (1.0, -0.5)], 16: [(0.0, 0.0), (3.0, -2.0)]}


@pytest.mark.parametrize("res,xi", [(r, a) for r in (8, 16) for a in ACTIONS[r]])
def test_shedding_statistics_match_the_oracle(envmod, oracle_mod, res, xi):
    state = _developed_wake(envmod, res)
    o = oracle_mod.Oracle(resolution=res, Re=100)
    o.set_state(*state)
    fo = np.empty((N_STEPS, 2), np.float32)
    for k in range(N_STEPS):
        fo[k] = o.update2(*xi)[0]
    so = _stats(fo, res)
    assert so[3] >= 3, "fewer than three shedding periods in the window: %r" % (so,)
    assert so[1] > 1e-3, "the wake does not shed: rms Cl %g" % so[1]
    for mode, name in ((0, "default (tree sums)"), (1, "perf mode (red-black)")):
        sg = _stats(_gpu_history(envmod, res, state, xi, mode), res)
        print("res %d xi %s %s: mean Cd %.6f / %.6f  rms Cl %.6f / %.6f  period %.2f / %.2f steps (gpu / oracle)"
              % (res, xi, name, sg[0], so[0], sg[1], so[1], sg[2], so[2]))
        assert abs(sg[0] - so[0]) <= TOL_CD * abs(so[0]), (name, sg, so)
        assert abs(sg[1] - so[1]) <= TOL_CL * so[1], (name, sg, so)
        assert abs(sg[2] - so[2]) <= TOL_ST * so[2], (name, sg, so)


def test_cfg4_wake_fixture_replays_on_the_oracle(envmod, oracle_mod):
    """tests/golden/wake_res32_re400.npz (the start state of the 512x256 bench line) was produced by the HIP path: four
    parity-mode steps from it, with and without actuation, are bit-identical on the oracle"""
    z = np.load(os.path.join(GOLDEN, "wake_res32_re400.npz"))
    g = envmod.AFCEnv(num_envs=2, resolution=32, reynolds=400, exact_sums=1)
    o = oracle_mod.Oracle(resolution=32, Re=400)
    g.set_reset_state(z["u"], z["v"], z["p"])
    g.reset()
    o.set_state(z["u"], z["v"], z["p"])
    for a in [(0.0, 0.0), (4.0, -3.0), (4.0, -3.0), (-1.5, 2.5)]:
        fg = g.update2(np.tile(np.array([a], np.float32), (2, 1)))
        fo, ito, _ = o.update2(*a)
        itg, _ = g.solver_stats()
        assert np.array_equal(fg[0], fo) and tuple(itg[0]) == tuple(ito)
    for x, y in zip(g.get_state(1), o.get_state()):
        assert np.array_equal(x, y)
    g.close()
