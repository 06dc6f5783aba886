"""PERF MODE (cfg.gs_mode = 1): MG.smooth's four Gauss-Seidel sweeps in red-black order (all four waves) instead of the
reference's lexicographic order (MG.pde:79-92).  A different smoother converges to the same pressure within the solver's
own tolerance (RMS residual 1e-4), not to the same iterates, so the comparison with the oracle is by tolerance -- the one
SURVEY.md section 7 states for this mode:
    |du|, |dv| <= 1e-3      |dp| <= 2e-2      |dCd|, |dCl| <= 3e-3        over 16 steps from init.bdim
and at four times that horizon the deviation must stay within 1.5 x of it (no drift of the force coefficients).  The headline benchmark and every parity test use gs_mode 0."""
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
    does (measured 3.7e-4 -> 7.7e-4 in u, 8.7e-3 -> 2.0e-2 in p from step 16 to step 64) -- at step 64 it must be inside
    1.5 x the tolerance stated for 16 steps, and the force coefficients must not drift apart"""
    dev = _deviation(envmod, oracle_mod, init_bdim, (3.5, -2.0), 64, (16, 64))
    d16, d64 = dev[16], dev[64]
    assert d16[0] <= TOL_U and d16[1] <= TOL_U and d16[2] <= TOL_P and d16[3] <= TOL_C, (d16, d64)
    assert d64[0] <= 1.5 * TOL_U and d64[1] <= 1.5 * TOL_U and d64[2] <= 1.5 * TOL_P and d64[3] <= TOL_C, (d16, d64)
    assert d64[3] <= 2.0 * max(d16[3], 1e-4), (d16, d64)  # max over all steps so far of |dC|: no late drift


def test_perf_mode_is_rejected_together_with_parity_mode(envmod):
    with pytest.raises(envmod.AfcError):
        envmod.AFCEnv(num_envs=1, resolution=8, reynolds=100, gs_mode=1, exact_sums=1)


# ---- the perf-mode solver against a red-black multigrid written out in numpy (float32, the reference's operation order) ----
def _f(x):
    return np.asarray(x, np.float32)


def _inv_of(lx, ly):
    """PoissonMatrix.pde:38-51: inv = -1/sumd where the diagonal does not vanish, else 1 (ghosts, solid interior)"""
    n, m = lx.shape
    inv = np.ones((n, m), np.float32)
    sumd = ((lx[1:-1, 1:-1] + lx[2:, 1:-1]) + ly[1:-1, 1:-1]) + ly[1:-1, 2:]
    with np.errstate(divide="ignore"):
        inv[1:-1, 1:-1] = np.where(sumd > np.float32(1e-5), np.float32(-1) / sumd, np.float32(1))
    return inv


def _times(lx, ly, x):
    """PoissonMatrix.times, PoissonMatrix.pde:53-64 (interior; ghosts of the result 0)"""
    out = np.zeros_like(x)
    lw, le, ls, ln = lx[1:-1, 1:-1], lx[2:, 1:-1], ly[1:-1, 1:-1], ly[1:-1, 2:]
    diag = -(((lw + le) + ls) + ln)
    out[1:-1, 1:-1] = (((x[1:-1, 1:-1] * diag + x[:-2, 1:-1] * lw) + x[2:, 1:-1] * le) + x[1:-1, :-2] * ls) + x[1:-1, 2:] * ln
    return out


def _setbc0(a):
    """Field.setBC of a pressure-type field (btype 0): ghosts mirror the adjacent interior cell, Field.pde:211-224"""
    a[0, :] = a[1, :]
    a[-1, :] = a[-2, :]
    a[:, 0] = a[:, 1]
    a[:, -1] = a[:, -2]


def _smooth_rb(lx, ly, inv, x, r, itmx):
    """MG.smooth (MG.pde:79-97) with the sweeps in RED-BLACK order: cells with (i + j) even first, then the odd ones;
    out-of-domain neighbours read as d = 0 during the sweeps (what the kernel's rings hold)"""
    n, m = r.shape
    d = (r * inv).astype(np.float32)
    ii, jj = np.meshgrid(np.arange(n), np.arange(m), indexing="ij")
    interior = (ii >= 1) & (ii <= n - 2) & (jj >= 1) & (jj <= m - 2)
    lw, le, ls, ln = lx, np.roll(lx, -1, 0), ly, np.roll(ly, -1, 1)
    for _ in range(itmx):
        for c in (0, 1):
            dz = d.copy()
            dz[~interior] = 0.0
            dW, dE, dS, dN = np.roll(dz, 1, 0), np.roll(dz, -1, 0), np.roll(dz, 1, 1), np.roll(dz, -1, 1)
            v = -((((dW * lw + dE * le) + dS * ls) + dN * ln) - r) * inv
            sel = interior & (((ii + jj) & 1) == c)
            d[sel] = v.astype(np.float32)[sel]
    d[~interior] = (r * inv).astype(np.float32)[~interior]
    _setbc0(d)
    x += d
    r -= _times(lx, ly, d)


def _restrict_field(r):
    n, m = r.shape
    nc, mc = (n - 2) // 2 + 2, (m - 2) // 2 + 2
    c = np.zeros((nc, mc), np.float32)
    c[1:-1, 1:-1] = ((r[1:-1:2, 1:-1:2] + r[1:-1:2, 2:-1:2]) + r[2:-1:2, 1:-1:2]) + r[2:-1:2, 2:-1:2]  # MG.pde:124-137
    _setbc0(c)
    return c


def _prolong(xc, n, m):
    i = np.arange(n)
    j = np.arange(m)
    ci = np.where(i < 1, 1, np.where(i > n - 2, n - 2, i))
    cj = np.where(j < 1, 1, np.where(j > m - 2, m - 2, j))
    d = xc[((ci - 1) // 2 + 1)[:, None], ((cj - 1) // 2 + 1)[None, :]].astype(np.float32)  # MG.pde:139-152
    return d


def _vcycle(levels, l, x, r):
    lx, ly, inv = levels[l]
    _smooth_rb(lx, ly, inv, x, r, 0)  # smooth(0): d = r*inv; d.setBC; increment  (MG.pde:69)
    rc = _restrict_field(r)
    xc = np.zeros_like(rc)
    if l + 1 < len(levels) - 1:
        _vcycle(levels, l + 1, xc, rc)
    lxc, lyc, invc = levels[l + 1]
    _smooth_rb(lxc, lyc, invc, xc, rc, 4)  # coarse.smooth(4), MG.pde:73
    d = _prolong(xc, *x.shape)
    x += d
    r -= _times(lx, ly, d)


@pytest.mark.parametrize("res,Re", [(8, 100), (24, 500)])
def test_perf_mode_solver_is_red_black_multigrid(envmod, res, Re):
    """MGsolver in perf mode (afc_op_mg_solve on a gs_mode = 1 env) against the same multigrid with red-black sweeps
    written out in numpy -- on the 128x64 hierarchy (even widths down to 4x2) and on the 384x192 one, whose coarsest
    levels are 14x8 -> 6x3 cells: ODD width, where one colour has a column more than the other (the
    kernel used to leave the last column of such a level unrelaxed).  The kernel sweeps strip by strip with the
    stages shifted by a row each, which is the same red-black iteration cell for cell, with the reference's operation
    order inside every update: the solutions are BIT-IDENTICAL, iteration counts equal.  (This test also caught an LDS
    overflow of the perf-mode rings on the wide grids: reads beyond a workgroup's LDS return 0, so the step-level
    tolerance tests still passed while the solver needed 16 V-cycles for what takes 2.)"""
    g = envmod.AFCEnv(num_envs=1, resolution=res, reynolds=Re, gs_mode=1)
    levels = []
    for l in range(g.num_levels):
        lx, ly = g.static("c.x", l), g.static("c.y", l)
        levels.append((lx, ly, _inv_of(lx, ly)))
    assert (levels[-1][0].shape[1] - 2) % 2 == (1 if res == 24 else 0)  # the odd-width coarsest level is in play
    # a right-hand side the singular Neumann problem can satisfy: b = A x* for a smooth x* (an arbitrary b is not in the
    # range of A -- the solid cells have no equation -- and 20 non-converging V-cycles only amplify round-off)
    lx0, ly0, _ = levels[0]
    ii, jj = np.meshgrid(np.arange(g.n, dtype=np.float32), np.arange(g.m, dtype=np.float32), indexing="ij")
    xs = (np.cos(3.0 * np.pi * ii / g.n) * np.cos(2.0 * np.pi * jj / g.m) + 0.3 * np.sin(5.0 * np.pi * ii / g.n)).astype(np.float32)
    _setbc0(xs)
    b = _times(lx0, ly0, xs)
    xg, itg, _ = g.op_mg_solve(np.zeros_like(b), b)
    # the reference loop (MG.pde:30-38, 46-66) around the red-black V-cycle
    x = np.zeros_like(b)
    lx, ly, inv = levels[0]
    r = (b - _times(lx, ly, x)).astype(np.float32)
    tol = np.float32(1e-8) * np.float32((g.n - 2) * (g.m - 2))
    it = 0
    while it < 20:
        it += 1
        _vcycle(levels, 0, x, r)
        _smooth_rb(lx, ly, inv, x, r, 4)
        if float(np.sum(r[1:-1, 1:-1].astype(np.float64) ** 2)) < tol:
            break
    assert itg == it and it < 20, (itg, it)
    scale = float(np.abs(x).max())
    print("perf-mode MG vs numpy red-black: res %d, %d iterations, max|dx| %.3g of max|x| %.3g" % (res, it, float(np.abs(xg - x).max()), scale))
    assert np.array_equal(xg, x), (float(np.abs(xg - x).max()), scale)
    g.close()


def test_register_resident_row_pipelines_equal_the_barrier_form_bit_for_bit(envmod, monkeypatch):
    """mg_up_rbw (round 5: the perf-mode up-leg as temporally blocked row pipelines out of registers, one per wave) runs
    the same half-sweeps on the same operands as the barrier form (gs_rb_stage): one and two V-cycles of the solver on
    the 128x64 and 256x128 hierarchies with EVERY level on the new path (AFC_RBW = mask of levels; the default is level
    0 only) against the barrier form everywhere, and 20 whole steps with the default."""
    def solve(res, mask, iters):
        monkeypatch.setenv("AFC_RBW", str(mask))
        g = envmod.AFCEnv(num_envs=1, resolution=res, reynolds=100, gs_mode=1, mg_max_iters=iters)
        rng = np.random.default_rng(3)
        b = np.zeros((g.n, g.m), np.float32)
        b[1:-1, 1:-1] = 1e-2 * rng.standard_normal((g.n - 2, g.m - 2)).astype(np.float32)
        b[1:-1, 1:-1] -= b[1:-1, 1:-1].mean()
        x, it, l = g.op_mg_solve(np.zeros_like(b), b)
        g.close()
        return x, it, l
    for res in (8, 16):
        for iters in (1, 2):
            x0, it0, l0 = solve(res, 0, iters)
            for mask in (1, 2, 4, 127):
                x1, it1, l1 = solve(res, mask, iters)
                assert it0 == it1 and np.array_equal(x0, x1), (res, iters, mask)

    def steps(mask):
        if mask is None:
            monkeypatch.delenv("AFC_RBW", raising=False)
        else:
            monkeypatch.setenv("AFC_RBW", str(mask))
        g = envmod.AFCEnv(num_envs=3, resolution=16, reynolds=100, gs_mode=1)
        xi = np.array([[0.0, 0.0], [3.0, -2.0], [-4.0, 1.0]], np.float32)
        f = [g.update2(xi).copy() for _ in range(20)]
        st = [g.get_state(e) for e in range(3)]
        it, _ = g.solver_stats()
        g.close()
        return f, st, it
    f0, s0, i0 = steps(0)
    f1, s1, i1 = steps(None)
    assert np.array_equal(i0, i1)
    for a, b in zip(f0, f1):
        assert np.array_equal(a, b)  # (r.r / sum(p) partial sums are doubles of float terms: the forces agree to the bit here)
    for (u0, v0, p0), (u1, v1, p1) in zip(s0, s1):
        assert np.array_equal(u0, u1) and np.array_equal(v0, v1) and np.array_equal(p0, p1)
