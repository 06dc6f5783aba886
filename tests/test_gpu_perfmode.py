"""PERF MODE (cfg.gs_mode = 1): MG.smooth's four Gauss-Seidel sweeps in red-black order (all four waves) instead of the
reference's lexicographic order (MG.pde:79-92).  A different smoother converges to the same pressure within the solver's
own tolerance (RMS residual 1e-4), not to the same iterates, so the comparison with the oracle is by tolerance -- the one
SURVEY.md section 7 states for this mode:
    |du|, |dv| <= 1e-3      |dp| <= 2e-2      |dCd|, |dCl| <= 3e-3        over 16 steps from init.bdim
and the deviation must not grow over 64 steps.  The headline benchmark and every parity test use gs_mode 0."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL_U, TOL_P, TOL_C = 1e-3, 2e-2, 3e-3
RES, RE = 24, 500  # the reference-native case: the only grid with a reference-authored state


@pytest.fixture(scope="module")
def envmod():
    from rlfluidcontrol_amd import build, env
    build.build()
    return env


def _deviation(envmod, oracle_mod, init, action, nsteps, checkpoints):
    g = envmod.AFCEnv(num_envs=2, resolution=RES, reynolds=RE, gs_mode=1)
    o = oracle_mod.Oracle(resolution=RES, Re=RE)
    g.set_reset_state(init["u"], init["v"], init["p"])
    g.reset()
    o.set_state(init["u"], init["v"], init["p"])
    xi = np.tile(np.asarray(action, np.float32)[None], (2, 1))
    out = {}
    dc = 0.0
    for k in range(1, nsteps + 1):
        fg = g.update2(xi)
        fo, _, _ = o.update2(float(action[0]), float(action[1]))
        assert np.array_equal(fg[0], fg[1])  # envs with identical inputs stay identical
        dc = max(dc, float(np.max(np.abs(fg[0].astype(np.float64) - fo)) * 2 / RES))
        if k in checkpoints:
            ug, vg, pg = g.get_state(0)
            uo, vo, po = o.get_state()
            out[k] = (float(np.abs(ug - uo).max()), float(np.abs(vg - vo).max()), float(np.abs(pg - po).max()), dc)
    itg, _ = g.solver_stats()
    assert int(itg.max()) <= 3  # the red-black smoother converges like the reference's (1 V-cycle, rarely 2)
    assert not g.diverged().any()
    g.close()
    return out


@pytest.mark.parametrize("action", [(0.0, 0.0), (3.5, -2.0)])
def test_perf_mode_within_the_stated_tolerance_over_16_steps(envmod, oracle_mod, init_bdim, action):
    dev = _deviation(envmod, oracle_mod, init_bdim, action, 16, (16,))[16]
    assert dev[0] <= TOL_U and dev[1] <= TOL_U and dev[2] <= TOL_P and dev[3] <= TOL_C, dev


def test_perf_mode_stays_within_the_tolerance_over_64_steps(envmod, oracle_mod, init_bdim):
    """four times the horizon the tolerance is stated for: the two runs are different (equally valid) discrete
    trajectories of a wake under constant actuation, so their difference grows at the rate any perturbation of this flow
    does (measured 6.5e-5 -> 7.9e-4 in u from step 16 to step 64) -- it must still be inside the stated tolerance, and
    the force coefficients must not drift apart"""
    dev = _deviation(envmod, oracle_mod, init_bdim, (3.5, -2.0), 64, (16, 64))
    d16, d64 = dev[16], dev[64]
    assert d64[0] <= TOL_U and d64[1] <= TOL_U and d64[2] <= TOL_P and d64[3] <= TOL_C, (d16, d64)
    assert d64[3] <= 2.0 * max(d16[3], 1e-4), (d16, d64)  # max over all steps so far of |dC|: no late drift


def test_perf_mode_is_rejected_together_with_parity_mode(envmod):
    with pytest.raises(envmod.AfcError):
        envmod.AFCEnv(num_envs=1, resolution=8, reynolds=100, gs_mode=1, exact_sums=1)
