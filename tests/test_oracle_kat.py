"""Oracle vs the analytic known-answer tests found in the reference's doc comments
(SURVEY.md §4, §8c item 2).  CPU only."""
import numpy as np
import pytest


def test_polygon_distance_kat(oracle_mod):
    # Body.pde:18-28: triangle (0.5,2.5),(2.75,0.25),(0,0): distance(3,1) "should equal sqrt(.5)"
    L = oracle_mod.lib()
    xs = np.array([0.5, 2.75, 0], np.float32)
    ys = np.array([2.5, 0.25, 0], np.float32)
    d = L.orc_polygon_distance(3, xs, ys, 0.0, 0.0, 3.0, 1.0)
    assert abs(d - np.sqrt(0.5)) < 2e-7


def test_poisson_matrix_kat(oracle_mod):
    # PoissonMatrix.pde:17-31: c == 1 (after setBC), x = i^2 - j^2  =>  A x == 0 in the interior
    L = oracle_mod.lib()
    n = 40
    lx = np.ones((n, n), np.float32)
    ly = np.ones((n, n), np.float32)
    L.orc_setbc(n, n, lx, 1, 0.0, 0)
    L.orc_setbc(n, n, ly, 2, 0.0, 0)
    i, j = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    x = (i * i - j * j).astype(np.float32)
    out = np.zeros_like(x)
    L.orc_poisson_times(n, n, lx, ly, x, out)
    assert np.all(out[2:-2, 2:-2] == 0.0)
    assert np.all(out[0, :] == 0) and np.all(out[:, 0] == 0)  # ghosts of the result are 0
    diag = np.zeros_like(x)
    inv = np.zeros_like(x)
    L.orc_poisson_diag_inv(n, n, lx, ly, diag, inv)
    assert np.all(inv[0, :] == 1) and np.all(inv[:, -1] == 1)  # PoissonMatrix.pde:43: inv defaults to 1
    assert np.all(diag[2:-2, 2:-2] == -4)
    assert np.allclose(inv[2:-2, 2:-2], -0.25)


def _mg_demo(n):
    # MG.pde:8-26 demo: u = (1,-0.5), u.x = 0 in [40,50)x[40,75), c == 1, b = div u
    ux = np.full((n, n), 1.0, np.float32)
    uy = np.full((n, n), -0.5, np.float32)
    ux[40:50, 40:75] = 0.0
    b = np.zeros((n, n), np.float32)
    b[1:-1, 1:-1] = ux[2:, 1:-1] - ux[1:-1, 1:-1] + uy[1:-1, 2:] - uy[1:-1, 1:-1]
    return b


def test_mg_demo_converges(oracle_mod):
    L = oracle_mod.lib()
    n = 130
    lx = np.ones((n, n), np.float32)
    ly = np.ones((n, n), np.float32)
    L.orc_setbc(n, n, lx, 1, 0.0, 0)
    L.orc_setbc(n, n, ly, 2, 0.0, 0)
    b = _mg_demo(n)
    x = np.zeros((n, n), np.float32)
    r = np.zeros_like(x)
    linf = []
    for _ in range(12):
        tol = L.orc_mg_updates(n, n, lx, ly, x, b, 1, r)
        linf.append(np.abs(r).max())
    # "note dependance on N": tol = sum_interior (1e-4)^2
    assert abs(tol - 1e-8 * 128 * 128) / (1e-8 * 128 * 128) < 1e-5
    assert linf[-1] < 1e-5  # MG.pde:25 stopping rule of the demo is reached
    assert linf[3] < linf[0] * 1e-2  # multigrid-rate convergence, not smoother-rate


def test_mg_hierarchy_depth(oracle_mod):
    # MG.pde:99-106: the coarsest level is the first that is not divisible (258x130 -> ... -> 6x4)
    def levels(n, m):
        k = 1
        while True:
            n, m = (n - 2) // 2 + 2, (m - 2) // 2 + 2
            k += 1
            if not ((n - 2) % 2 == 0 and (m - 2) % 2 == 0 and n > 4 and m > 4):
                return k, (n, m)
    assert levels(130, 66) == (6, (6, 4))
    assert levels(258, 130) == (7, (6, 4))
    assert levels(386, 194) == (7, (8, 5))
    assert levels(514, 258) == (8, (6, 4))


def test_setbc_order(oracle_mod):
    # Field.pde:209-234 on a random btype-1 gradientExit field
    L = oracle_mod.lib()
    rng = np.random.default_rng(0)
    n, m = 12, 9
    a = rng.standard_normal((n, m)).astype(np.float32)
    ref = a.copy()
    L.orc_setbc(n, m, a, 1, 1.0, 1)
    assert np.all(a[1, :] == 1.0)
    # ghost copy happened BEFORE a[1][j] = bval, except row j=0 / m-1 which the y-loop overwrote
    assert np.all(a[0, 1:-1] == ref[1, 1:-1])
    s = np.float32(0)
    for j in range(1, m - 1):
        s += ref[n - 2, j]
    s = np.float32(s / np.float32(m - 2))
    exp = (ref[n - 2, 1:-1] + (np.float32(1.0) - s)).astype(np.float32)
    assert np.array_equal(a[n - 1, 1:-1], exp)
    # corners: y-loop copies after the x-loop
    assert a[0, 0] == a[0, 1]
    # outflow corner is copied before the mean correction is applied to j=1..m-2
    assert a[n - 1, m - 1] == ref[n - 2, m - 2]
    b = rng.standard_normal((n, m)).astype(np.float32)
    L.orc_setbc(n, m, b, 2, 0.0, 0)
    assert np.all(b[:, 1] == 0) and np.all(b[:, m - 1] == 0) and np.all(b[0, 2:-1] == b[1, 2:-1])


def test_delta_kernels(oracle_mod):
    L = oracle_mod.lib()
    assert L.orc_delta0(-2.0) == 0 and L.orc_delta0(2.0) == 1 and abs(L.orc_delta0(0.0) - 0.5) < 1e-7
    assert L.orc_delta1(2.0) == 0 and L.orc_delta1(-2.5) == 0
    # delta1(0) = 0.25*eps - (1/2pi)*(eps/pi*2) = 0.5 - 2/pi^2
    assert abs(L.orc_delta1(0.0) - (0.5 - 2 / np.pi ** 2)) < 1e-6
    assert L.orc_union_delta0(-1.0) == 0 and L.orc_union_delta0(1.0) == 1


def test_linear_clamp_and_stagger(oracle_mod):
    L = oracle_mod.lib()
    n, m = 6, 5
    a = np.arange(n * m, dtype=np.float32).reshape(n, m)
    assert L.orc_linear(n, m, a, 0, 2.0, 3.0) == a[2, 3]
    assert abs(L.orc_linear(n, m, a, 0, 2.5, 3.0) - 0.5 * (a[2, 3] + a[3, 3])) < 1e-6
    assert L.orc_linear(n, m, a, 0, -5.0, 1.0) == pytest.approx(0.5 * (a[0, 1] + a[1, 1]))
    assert L.orc_linear(n, m, a, 1, 1.5, 2.0) == a[2, 2]  # btype 1: x += 0.5
