"""GPU parity: the HIP path (through the C ABI) vs the CPU oracle on identical inputs.

Tolerances.  The HIP step keeps the reference's float32 operation order (no FMA, lexicographic
Gauss-Seidel order, sequential outflow mean).  With cfg.exact_sums=1 (parity mode) the two
remaining order-sensitive reductions (Field.sum for the mean pressure, Field.inner for the MG
stopping test) are also accumulated sequentially and the WHOLE STEP IS REQUIRED TO BE
BIT-IDENTICAL to the oracle: u, v, p, forces, iteration counts, over 16 steps.
In the default mode those two sums are trees; the resulting last-bit change of the uniform
pressure shift perturbs later solves at round-off level, and since p is only solved to the
reference's tolerance (RMS residual 1e-4, one V-cycle) the stated tolerance for <= 16 steps is
    |du|,|dv| <= 5e-6   |dp| <= 5e-4*max(1,max|p|)   |dCd|,|dCl| <= 5e-5*max(1,|C|)
(measured: du <= 1.6e-6, dp <= 3.4e-4 with max|p| ~ 1e2 at cold start, dC <= 1.4e-4 on C ~ 200).
Operators with no such reduction in them are required to match bit for bit in both modes.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL_U, TOL_P, TOL_F = 5e-6, 5e-4, 5e-5


@pytest.fixture(scope="module")
def envmod():
    from rlfluidcontrol_amd import build, env
    build.build()
    return env


def _close(a, b, tol, rel=False):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    scale = max(1.0, float(np.max(np.abs(b)))) if rel else 1.0
    return float(np.max(np.abs(a - b))) <= tol * scale


@pytest.mark.parametrize("res,Re", [(8, 100), (16, 100), (24, 500)])
def test_static_fields_bit_exact(envmod, oracle_mod, res, Re):
    """host precompute of the product vs BDIM.get_coeffs of the oracle (incl. wall-normal ties)"""
    g = envmod.AFCEnv(num_envs=1, resolution=res, reynolds=Re)
    o = oracle_mod.Oracle(resolution=res, Re=Re)
    for name in ["del.x", "del.y", "del1.x", "del1.y", "wnx.x", "wnx.y", "wny.x", "wny.y"]:
        assert np.array_equal(g.static(name), o.coeff(name)), name
    # c = del*dt is only materialised by BDIM.update(): take it after one oracle step
    o.update2(0.0, 0.0)
    assert np.array_equal(g.static("c.x"), o.coeff("c.x"))
    assert np.array_equal(g.static("c.y"), o.coeff("c.y"))
    # ub through the weights: compare after a get_coeffs with known dphi
    o.refresh_coeffs(0.3, -0.2)
    ubx = o.coeff("ub.x")
    w1, w2 = g.static("w1.x"), g.static("w2.x")
    assert np.count_nonzero(ubx) > 0
    assert np.all((ubx != 0) <= ((w1 != 0) | (w2 != 0)))
    g.close()


def _wake_like(envmod, oracle_mod, res, Re, init_bdim):
    """a velocity / pressure state with structure on the (n, m) grid of this resolution: the reference's checkpoint
    where it fits, else a few oracle steps from the free stream under a strong action"""
    o = oracle_mod.Oracle(resolution=res, Re=Re)
    if init_bdim["u"].shape == (o.n, o.m):
        return init_bdim["u"].copy(), init_bdim["v"].copy(), init_bdim["p"].copy()
    for _ in range(3):
        o.update2(4.0, -3.0)
    return o.get_state()


@pytest.mark.parametrize("impl", [0, 2])
@pytest.mark.parametrize("same", [False, True])
@pytest.mark.parametrize("res,Re", [(8, 100), (16, 100), (24, 500), (32, 400)])
def test_advdif_bit_exact(envmod, oracle_mod, init_bdim, res, Re, same, impl):
    """VectorField.AdvDif alone, through the form of the pass the step itself would run on this grid (the row
    pipelines advdif_v3 where they apply, the LDS tiles advdif_v2 elsewhere); same: u0 = u (the predictor's form)"""
    if impl == 0 and same:
        pytest.skip("the straightforward kernels have one form")
    g = envmod.AFCEnv(num_envs=1, resolution=res, reynolds=Re, impl=impl)
    L = oracle_mod.lib()
    u, v, _ = _wake_like(envmod, oracle_mod, res, Re, init_bdim)
    rng = np.random.default_rng(1)
    if same:
        u0, v0 = u.copy(), v.copy()
        fx, fy = g.op_advdif(u, v)
    else:
        u0 = (u + 0.01 * rng.standard_normal(u.shape)).astype(np.float32)
        v0 = (v + 0.01 * rng.standard_normal(v.shape)).astype(np.float32)
        fx, fy = g.op_advdif(u, v, u0, v0)
    ox, oy = u.copy(), v.copy()
    L.orc_advdif(g.n, g.m, ox, oy, u0, v0, g.dt, np.float32(res) / np.float32(Re))
    assert np.array_equal(fx, ox) and np.array_equal(fy, oy)
    g.close()


def _seq_sum_interior(a):
    """Field.sum (Field.pde:311-318): sequential float32 sum over the interior, i outer, j inner"""
    return np.cumsum(a[1:-1, 1:-1].reshape(-1), dtype=np.float32)[-1]


@pytest.mark.parametrize("res,Re", [(8, 100), (16, 100), (24, 500), (32, 400)])
def test_project_bit_exact(envmod, oracle_mod, init_bdim, res, Re):
    """VectorField.project alone (VectorField.pde:130-143), through the fused projection the step runs on this grid:
    solve head (head_v3 / mg_down_fused), coarse down-legs (down_v3c), the LDS-resident bottom, up-legs, projection
    tail -- against the oracle's MGsolver on b = div u and the reference's own follow-up arithmetic in numpy float32"""
    import ctypes as C
    g = envmod.AFCEnv(num_envs=1, resolution=res, reynolds=Re, exact_sums=1)
    o = oracle_mod.Oracle(resolution=res, Re=Re)
    u, v, p = _wake_like(envmod, oracle_mod, res, Re, init_bdim)
    rng = np.random.default_rng(7)
    u = u.copy()
    v = v.copy()
    u[1:-1, 1:-1] += (0.02 * rng.standard_normal((g.n - 2, g.m - 2))).astype(np.float32)  # a divergent field
    v[1:-1, 1:-1] += (0.02 * rng.standard_normal((g.n - 2, g.m - 2))).astype(np.float32)
    L = oracle_mod.lib()
    L.orc_setbc(g.n, g.m, u, 1, 1.0, 1)
    L.orc_setbc(g.n, g.m, v, 2, 0.0, 0)
    o2 = oracle_mod.Oracle(resolution=res, Re=Re)
    o2.update2(0.0, 0.0)
    cx, cy = o2.coeff("c.x"), o2.coeff("c.y")
    gu, gv, gp, itg, lg = g.op_project(u, v, p)
    # the reference, statement by statement
    b = np.zeros_like(u)
    b[1:-1, 1:-1] = ((u[2:, 1:-1] - u[1:-1, 1:-1]) + v[1:-1, 2:]) - v[1:-1, 1:-1]  # VectorField.pde:56-65
    po = p.copy()
    lo = C.c_float()
    ito = L.orc_mg_solve(g.n, g.m, cx, cy, po, b, 20, C.byref(lo))
    s = np.float32(-1) * _seq_sum_interior(po) / np.float32((g.n - 2) * (g.m - 2))  # VectorField.pde:136
    po = (po + s).astype(np.float32)
    dpx = np.zeros_like(po)
    dpy = np.zeros_like(po)
    dpx[1:-1, 1:-1] = po[1:-1, 1:-1] - po[:-2, 1:-1]  # Field.pde:71-81
    dpy[1:-1, 1:-1] = po[1:-1, 1:-1] - po[1:-1, :-2]
    L.orc_setbc(g.n, g.m, dpx, 1, 0.0, 0)
    L.orc_setbc(g.n, g.m, dpy, 2, 0.0, 0)
    uo = (u + cx * (dpx * np.float32(-1))).astype(np.float32)  # VectorField.pde:138: plusEq(coeff.times(dp.times(-1)))
    vo = (v + cy * (dpy * np.float32(-1))).astype(np.float32)
    L.orc_setbc(g.n, g.m, uo, 1, 1.0, 1)
    L.orc_setbc(g.n, g.m, vo, 2, 0.0, 0)
    assert itg == ito
    assert np.array_equal(gp, po)
    assert np.array_equal(gu, uo) and np.array_equal(gv, vo)
    g.close()


def test_setbc_u(envmod, oracle_mod):
    g = envmod.AFCEnv(num_envs=1, resolution=8, reynolds=100)
    L = oracle_mod.lib()
    rng = np.random.default_rng(2)
    ux = (1 + 0.1 * rng.standard_normal((g.n, g.m))).astype(np.float32)
    uy = (0.1 * rng.standard_normal((g.n, g.m))).astype(np.float32)
    gx, gy = g.op_setbc_u(ux, uy)
    ox, oy = ux.copy(), uy.copy()
    L.orc_setbc(g.n, g.m, ox, 1, 1.0, 1)
    L.orc_setbc(g.n, g.m, oy, 2, 0.0, 0)
    assert np.array_equal(gy, oy)
    assert np.array_equal(gx, ox)  # the outflow mean is summed in the reference's sequential order
    g.close()


@pytest.mark.parametrize("impl", [0, 2])
@pytest.mark.parametrize("res,Re", [(8, 100), (16, 100), (24, 500), (32, 400)])
def test_mg_solver_vs_oracle(envmod, oracle_mod, res, Re, impl):
    """MGsolver on the env's own coefficient hierarchy, random divergence-like rhs"""
    g = envmod.AFCEnv(num_envs=1, resolution=res, reynolds=Re, impl=impl)
    o = oracle_mod.Oracle(resolution=res, Re=Re)
    o.update2(0.0, 0.0)
    lx, ly = o.coeff("c.x"), o.coeff("c.y")
    rng = np.random.default_rng(3)
    b = np.zeros((g.n, g.m), np.float32)
    b[1:-1, 1:-1] = 1e-2 * rng.standard_normal((g.n - 2, g.m - 2)).astype(np.float32)
    b[1:-1, 1:-1] -= b[1:-1, 1:-1].mean()
    x0 = np.zeros_like(b)
    xg, itg, lg = g.op_mg_solve(x0, b)
    xo = x0.copy()
    import ctypes as C
    lo = C.c_float()
    ito = oracle_mod.lib().orc_mg_solve(g.n, g.m, lx, ly, xo, b, 20, C.byref(lo))
    assert itg == ito
    # no reduction feeds back into x within one solve except the stopping test -> bit exact
    assert np.array_equal(xg, xo)
    assert abs(lg - lo.value) <= 1e-9
    g.close()


def _run_pair(envmod, oracle_mod, res, Re, nsteps, actions, init=None, num_envs=1, exact=0, impl=2, fuse_tail=0, upleg=0):
    g = envmod.AFCEnv(num_envs=num_envs, resolution=res, reynolds=Re, exact_sums=exact, impl=impl, fuse_tail=fuse_tail,
                      upleg=upleg)
    o = oracle_mod.Oracle(resolution=res, Re=Re)
    if init is not None:
        g.set_reset_state(init["u"], init["v"], init["p"])
        o.set_state(init["u"], init["v"], init["p"])
    out = []
    for k in range(nsteps):
        a = actions[k]
        xi = np.tile(np.asarray(a, np.float32)[None], (num_envs, 1))
        fg = g.update2(xi)
        fo, ito, _ = o.update2(float(a[0]), float(a[1]))
        itg, _ = g.solver_stats()
        out.append((fg, fo, itg, ito))
    return g, o, out


def test_step_parity_native_from_init_bdim(envmod, oracle_mod, init_bdim):
    """reference-native case (384x192, Re 500) continued from the reference's checkpoint"""
    acts = [(0, 0)] * 2 + [(3.5, -2.0)] * 6
    g, o, out = _run_pair(envmod, oracle_mod, 24, 500, len(acts), acts, init=init_bdim)
    for fg, fo, itg, ito in out:
        assert tuple(itg[0]) == tuple(ito)
        assert _close(fg[0] * 2 / 24, fo * 2 / 24, TOL_F, rel=True)
    ug, vg, pg = g.get_state()
    uo, vo, po = o.get_state()
    assert _close(ug, uo, TOL_U) and _close(vg, vo, TOL_U) and _close(pg, po, TOL_P, rel=True)
    assert _close(g.vorticity(), o.vorticity(), 2 * TOL_U)
    assert _close(g.probes(32)[0], o.probes(32), TOL_P, rel=True)
    assert abs(g.t[0] - o.t) < 1e-7
    g.close()


@pytest.mark.parametrize("res,Re", [(8, 100), (16, 100)])
def test_step_parity_cold_start(envmod, oracle_mod, res, Re):
    """cfg1/cfg2 grids from the free stream, 16 steps with a saturated then mixed action"""
    acts = [(0, 0)] * 4 + [(5.0, -5.0)] * 6 + [(-2.5, 1.0)] * 6
    g, o, out = _run_pair(envmod, oracle_mod, res, Re, len(acts), acts, num_envs=3)
    for fg, fo, itg, ito in out:
        assert tuple(itg[0]) == tuple(ito)
        for e in range(3):  # envs with identical inputs stay identical
            assert np.array_equal(fg[e], fg[0])
        assert _close(fg[0] * 2 / res, fo * 2 / res, TOL_F, rel=True)
    for e in (0, 2):
        ug, vg, pg = g.get_state(e)
        uo, vo, po = o.get_state()
        assert _close(ug, uo, TOL_U) and _close(vg, vo, TOL_U) and _close(pg, po, TOL_P, rel=True)
    g.close()


def test_envs_are_independent(envmod, oracle_mod):
    """different actions per env: each env matches its own oracle"""
    res, Re, E = 8, 100, 4
    g = envmod.AFCEnv(num_envs=E, resolution=res, reynolds=Re)
    rng = np.random.default_rng(5)
    acts = (5 * rng.uniform(-1, 1, size=(6, E, 2))).astype(np.float32)
    oracles = [oracle_mod.Oracle(resolution=res, Re=Re) for _ in range(E)]
    for k in range(6):
        fg = g.update2(acts[k])
        for e, o in enumerate(oracles):
            fo, _, _ = o.update2(float(acts[k, e, 0]), float(acts[k, e, 1]))
            assert _close(fg[e] * 2 / res, fo * 2 / res, TOL_F, rel=True)
    for e, o in enumerate(oracles):
        ug, vg, pg = g.get_state(e)
        uo, vo, po = o.get_state()
        assert _close(ug, uo, TOL_U) and _close(vg, vo, TOL_U) and _close(pg, po, TOL_P, rel=True)
    g.close()


@pytest.mark.parametrize("impl", [0, 2])
@pytest.mark.parametrize("res,Re,init", [(8, 100, False), (16, 100, False), (24, 500, True), (32, 400, False)])
def test_parity_mode_is_bit_exact(envmod, oracle_mod, init_bdim, res, Re, init, impl):
    """cfg.exact_sums=1: every float the step produces equals the oracle's, bit for bit."""
    acts = [(0, 0)] * 3 + [(5.0, -5.0)] * 5 + [(-2.5, 1.0)] * 4 + [(0, 0)] * 4
    g, o, out = _run_pair(envmod, oracle_mod, res, Re, len(acts), acts, init=init_bdim if init else None, exact=1,
                          impl=impl)
    for k, (fg, fo, itg, ito) in enumerate(out):
        assert tuple(itg[0]) == tuple(ito), k
        assert np.array_equal(fg[0], fo), (k, fg[0], fo)
    ug, vg, pg = g.get_state()
    uo, vo, po = o.get_state()
    assert np.array_equal(ug, uo) and np.array_equal(vg, vo) and np.array_equal(pg, po)
    assert np.array_equal(g.vorticity(), o.vorticity())
    assert np.array_equal(g.probes(32)[0], o.probes(32))
    _, rg = g.solver_stats()
    g.close()


@pytest.mark.parametrize("upleg", [1, 2, 3])
@pytest.mark.parametrize("res,Re", [(8, 100), (16, 100)])
def test_each_form_of_the_upleg_is_bit_exact(envmod, oracle_mod, res, Re, upleg):
    """afc_config.upleg = 1 (strip pipeline, afc::prod), 2 (barrier form, afc::diag), 3 (march down the rows, afc::lat): each
    kernel instantiation on its own against the oracle in parity mode -- MG.smooth's sweeps (MG.pde:79-92) in the reference's
    order, so every float and every iteration count equals the oracle's (cold start: the first solves take several V-cycles,
    the replay of the finest level included)."""
    acts = [(0, 0)] * 3 + [(5.0, -5.0)] * 5 + [(-2.5, 1.0)] * 4
    g, o, out = _run_pair(envmod, oracle_mod, res, Re, len(acts), acts, exact=1, upleg=upleg)
    assert g.upleg() == upleg
    for k, (fg, fo, itg, ito) in enumerate(out):
        assert tuple(itg[0]) == tuple(ito), k
        assert np.array_equal(fg[0], fo), (k, fg[0], fo)
    ug, vg, pg = g.get_state()
    uo, vo, po = o.get_state()
    assert np.array_equal(ug, uo) and np.array_equal(vg, vo) and np.array_equal(pg, po)
    g.close()


@pytest.mark.parametrize("res,Re", [(8, 100), (16, 100)])
def test_fused_projection_tail_is_bit_exact(envmod, oracle_mod, res, Re):
    """afc_config.fuse_tail = 1: the predictor's projection tail (VectorField.pde:136-140) and the row part of setBC run
    inside the corrector's advection pass (advdif_v3f) -- in parity mode every float still equals the oracle's; and with
    the default sums the fused and the two-pass form agree bit for bit with each other (from a cold start: the first
    steps take several V-cycles, saturated and mixed actions, several envs)."""
    acts = [(0, 0)] * 3 + [(5.0, -5.0)] * 5 + [(-2.5, 1.0)] * 4 + [(0, 0)] * 4
    g, o, out = _run_pair(envmod, oracle_mod, res, Re, len(acts), acts, exact=1, fuse_tail=1)
    assert g.fused_tail() is True
    for k, (fg, fo, itg, ito) in enumerate(out):
        assert tuple(itg[0]) == tuple(ito), k
        assert np.array_equal(fg[0], fo), (k, fg[0], fo)
    ug, vg, pg = g.get_state()
    uo, vo, po = o.get_state()
    assert np.array_equal(ug, uo) and np.array_equal(vg, vo) and np.array_equal(pg, po)
    g.close()
    E = 5
    ga = envmod.AFCEnv(num_envs=E, resolution=res, reynolds=Re, fuse_tail=1)
    gb = envmod.AFCEnv(num_envs=E, resolution=res, reynolds=Re, fuse_tail=-1)
    assert ga.fused_tail() is True and gb.fused_tail() is False
    rng = np.random.default_rng(23)
    a = (5 * rng.uniform(-1, 1, size=(10, E, 2))).astype(np.float32)
    for k in range(10):
        assert np.array_equal(ga.update2(a[k]), gb.update2(a[k])), k
    for e in range(E):
        for x, y in zip(ga.get_state(e), gb.get_state(e)):
            assert np.array_equal(x, y), e
    ga.close()
    gb.close()


def test_fused_projection_tail_is_chosen_by_occupancy(envmod):
    """default (fuse_tail = 0): fused where a launch has at least one env per CU and the grid is one column tile wide"""
    import torch
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    g = envmod.AFCEnv(num_envs=4, resolution=8, reynolds=100)
    assert g.fused_tail() is False
    g.close()
    g = envmod.AFCEnv(num_envs=cus, resolution=8, reynolds=100)
    assert g.fused_tail() is True
    g.close()
    g = envmod.AFCEnv(num_envs=2, resolution=24, reynolds=500, fuse_tail=1)  # two column tiles: not available
    assert g.fused_tail() is False
    g.close()


@pytest.mark.parametrize("res,Re", [(8, 100), (16, 100)])
def test_lazy_residual_writeback_is_invisible(envmod, res, Re):
    """Default mode does not write the finest-level residual back unless a second V-cycle needs it
    (then the level is replayed).  Same fields, bit for bit, as always writing it (dbg switch 2),
    on a cold start whose first steps do take several V-cycles."""
    E = 2
    ga = envmod.AFCEnv(num_envs=E, resolution=res, reynolds=Re)
    gb = envmod.AFCEnv(num_envs=E, resolution=res, reynolds=Re, dbg=2)
    rng = np.random.default_rng(11)
    acts = (5 * rng.uniform(-1, 1, size=(12, E, 2))).astype(np.float32)
    max_it = 0
    for k in range(12):
        fa = ga.update2(acts[k])
        fb = gb.update2(acts[k])
        assert np.array_equal(fa, fb), k
        ita, _ = ga.solver_stats()
        itb, _ = gb.solver_stats()
        assert np.array_equal(ita, itb)
        max_it = max(max_it, int(ita.max()))
    assert max_it >= 2, "the run never needed a second V-cycle: replay path not exercised"
    for e in range(E):
        for a, b in zip(ga.get_state(e), gb.get_state(e)):
            assert np.array_equal(a, b)
    ga.close()
    gb.close()


def test_full_occupancy_is_deterministic_and_env_independent(envmod):
    """1024 envs (four workgroups per CU, the benchmark's occupancy) stepping from the same state with the
    same actions must stay bit-identical to each other and to a 1-env run: no workgroup may see another
    one's data, and nothing in the step may depend on timing (relaxed LDS-only barriers included)."""
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "wake_res16_re100.npz"))
    outs = []
    for E in (1024, 1):
        g = envmod.AFCEnv(num_envs=E, resolution=16, reynolds=100)
        g.set_reset_state(z["u"], z["v"], z["p"])
        g.reset()
        rng = np.random.default_rng(3)
        obs_all = []
        for k in range(3):
            a = rng.uniform(-1, 1, (1, 2)).astype(np.float32)
            obs, rew, done, ns = g.agent_step(np.tile(a, (E, 1)))
            assert (ns == ns[0]).all()
            assert (obs == obs[0]).all() and (rew == rew[0]).all(), k
            obs_all.append(obs[0].copy())
        state = [g.get_state(e) for e in ((0, E // 2, E - 1) if E > 1 else (0,))]
        for st in state[1:]:
            for x, y in zip(state[0], st):
                assert np.array_equal(x, y)
        outs.append((np.array(obs_all), state[0]))
        g.close()
    assert np.array_equal(outs[0][0], outs[1][0])
    for x, y in zip(outs[0][1], outs[1][1]):
        assert np.array_equal(x, y)
