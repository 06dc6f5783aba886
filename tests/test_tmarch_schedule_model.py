"""The schedule of mg_up_tr (rlfluidcontrol_amd/csrc/bdim_v2_tmarch.h) on the CPU: one wave takes a whole multigrid level in
one pass DOWN THE ROWS -- lane 16 k + c owns the CB columns 1 + CB c .. CB (c + 1) of sweep k and at step T stands on row
T - c - 3 k -- fetching (i+1, own columns), (i, first column of the next block) and r one step ahead, storing in place into
rings of NR rows whose slots are taken by rows g and g + NR of the same column block, with rows entering and leaving
along diagonals under the three counters F (filled), M (march steps done), X (exited).

This model executes exactly that schedule in numpy float32 -- the lanes of a step in SIMD fashion (all reads of a step
before its writes), fills as EARLY as the exits allow and as LATE as the march tolerates -- and checks
  * every cell gets the bits of the reference's sequential loops (MG.pde:82-89: four in-place lexicographic sweeps);
  * no fill overwrites a ring cell that a pending exit (its own diagonal or the next) still has to read;
  * the march never reads a ring slot that holds another row than the one it means;
  * no wait of the protocol points forwards (a fill that needs an exit which needs a march step which needs that fill).
The GPU side of the same statement: tools/upleg_bench.hip (bit comparison with mg_up_smooth on random data incl. rows that
cross a body), tests/test_gpu_round2.py::test_uplegs_agree and the parity suite."""
import numpy as np
import pytest

f32 = np.float32


def reference(r, inv, lx, ly, n_, m_):
    d = np.zeros((n_ + 2, m_ + 2), f32)
    d[1:-1, 1:-1] = (r[1:-1, 1:-1] * inv[1:-1, 1:-1]).astype(f32)  # MG.pde:80
    for _ in range(4):
        for i in range(1, n_ + 1):
            for j in range(1, m_ + 1):
                t = f32(f32(d[i - 1, j] * lx[i, j]) + f32(d[i + 1, j] * lx[i + 1, j]))
                t = f32(t + f32(d[i, j - 1] * ly[i, j]))
                t = f32(t + f32(d[i, j + 1] * ly[i, j + 1]))
                t = f32(t - r[i, j])
                d[i, j] = f32(f32(-t) * inv[i, j])
    return d


def schedule(r, inv, lx, ly, n_, m_, CB, NR, early, LG=3):
    SPAN = 15 + 3 * LG
    KX = 11 + (1 if CB == 2 else 0)
    Dmax = n_ + 15
    ringd = np.full((NR, m_ + 2), np.nan, f32)
    ringr = np.full((NR, m_ + 2), np.nan, f32)
    owner = np.full((NR, 16), -10 ** 9)  # the row a (slot, column block) holds
    out = np.zeros((n_ + 2, m_ + 2), f32)
    st = dict(F=0, M=0, X=0)
    Tend = n_ + SPAN

    def fill_diag(D):
        for c in range(16):
            g = D - c
            if 1 <= g <= n_:
                s = g % NR
                prev = owner[s, c]
                if prev > 0:  # the exits of diagonal prev + c and of the next one have read it
                    assert prev + c + 1 <= st["X"], ("fill overwrites a live ring cell", g, c, prev, st)
                owner[s, c] = g
                j0 = 1 + CB * c
                ringr[s, j0:j0 + CB] = r[g, j0:j0 + CB]
                ringd[s, j0:j0 + CB] = (r[g, j0:j0 + CB] * inv[g, j0:j0 + CB]).astype(f32)
                if c == 0:
                    ringd[s, 0] = 0
                if c == 15:
                    ringd[s, m_ + 1] = 0

    def exit_diag(D):
        for c in range(16):
            g = D - c
            if 1 <= g <= n_:
                s = g % NR
                assert owner[s, c] == g
                j0 = 1 + CB * c
                out[g, j0:j0 + CB] = ringd[s, j0:j0 + CB]

    def rd(ring, g, j, blk):
        if g < 1 or g > n_:
            return f32(0)
        s = g % NR
        assert owner[s, blk] == g, ("the march reads another row's slot", g, blk, owner[s, blk])
        v = ring[s, j]
        assert not np.isnan(v)
        return v

    up = np.zeros((64, CB), f32)
    Rr = np.zeros((64, CB), f32)

    def fetch(T):
        Dn = np.zeros((64, CB), f32)
        Xn = np.zeros(64, f32)
        Rn = np.zeros((64, CB), f32)
        for lane in range(64):
            k, c = lane >> 4, lane & 15
            j0, i = 1 + CB * c, T - c - LG * k
            for e in range(CB):
                Dn[lane, e] = rd(ringd, i + 1, j0 + e, c)
                Rn[lane, e] = rd(ringr, i, j0 + e, c)
            if 1 <= i <= n_:
                Xn[lane] = rd(ringd, i, j0 + CB, c + 1) if c < 15 else ringd[i % NR, m_ + 1]
        return Dn, Xn, Rn

    def step(T, buf):
        Dn, Xn, Rn = buf
        newup = up.copy()
        stores = []
        for lane in range(64):
            k, c = lane >> 4, lane & 15
            j0, i = 1 + CB * c, T - c - LG * k
            L = up[lane - 1, CB - 1] if c > 0 else f32(0)  # DPP row_shr:1
            nw = np.zeros(CB, f32)
            gi, gi1 = min(max(i, 0), n_ + 1), min(max(i + 1, 0), n_ + 1)
            for e in range(CB):
                Rg = Rr[lane, e] if e < CB - 1 else Xn[lane]
                j = j0 + e
                t = f32(f32(up[lane, e] * lx[gi, j]) + f32(Dn[lane, e] * lx[gi1, j]))
                t = f32(t + f32(L * ly[gi, j]))
                t = f32(t + f32(Rg * ly[gi, j + 1]))
                t = f32(t - Rn[lane, e])
                nw[e] = f32(f32(-t) * inv[gi, j])
                L = nw[e]
            if 1 <= i <= n_:
                stores.append((i, j0, nw.copy(), c))
                newup[lane] = nw
            for e in range(CB - 1):
                Rr[lane, e] = Dn[lane, e + 1]
        for i, j0, nw, c in stores:  # all reads of a step precede its writes
            assert owner[i % NR, c] == i
            ringd[i % NR, j0:j0 + CB] = nw
        up[:] = newup

    def helpers():
        go = True
        while go:
            go = False
            if st["X"] < Dmax and st["M"] >= min(st["X"] + 1 + KX, Tend + 2):
                exit_diag(st["X"] + 1)
                st["X"] += 1
                go = True
            if early and st["F"] < Dmax and st["X"] >= min(st["F"] + 1 - NR + 1, Dmax):
                fill_diag(st["F"] + 1)
                st["F"] += 1
                go = True

    buf = None
    for T in range(0, Tend + 1, 2):
        need = min(T + 3, Dmax)
        helpers()
        while st["F"] < need:  # as late as the march tolerates
            assert st["X"] >= min(st["F"] + 1 - NR + 1, Dmax), ("a wait that points forwards", st)
            fill_diag(st["F"] + 1)
            st["F"] += 1
        if T == 0:
            buf = fetch(0)
        b1 = fetch(T + 1)
        step(T, buf)
        b2 = fetch(T + 2)
        step(T + 1, b1)
        buf = b2
        st["M"] = T + 2
    helpers()
    while st["X"] < Dmax:
        assert st["M"] >= min(st["X"] + 1 + KX, Tend + 2)
        exit_diag(st["X"] + 1)
        st["X"] += 1
    return out


def _case(CB, n_, seed=1):
    m_ = 16 * CB
    rng = np.random.default_rng(seed)
    r = ((rng.random((n_ + 2, m_ + 2), dtype=f32) - f32(0.5)) * f32(0.01)).astype(f32)
    lx = np.full((n_ + 3, m_ + 3), 0.25, f32)
    ly = np.full((n_ + 3, m_ + 3), 0.25, f32)
    bi0, bi1, bj0, bj1 = n_ // 4 - 2, n_ // 4 + 3, m_ // 2 - m_ // 9, m_ // 2 + m_ // 9  # rows that cross a body
    lx[bi0:bi1 + 1, bj0:bj1 + 1] = (0.25 * (0.25 + 0.5 * rng.random((bi1 - bi0 + 1, bj1 - bj0 + 1)))).astype(f32)
    ly[bi0:bi1 + 1, bj0:bj1 + 1] = (0.25 * (0.25 + 0.5 * rng.random((bi1 - bi0 + 1, bj1 - bj0 + 1)))).astype(f32)
    inv = np.ones((n_ + 2, m_ + 2), f32)
    for i in range(1, n_ + 1):
        for j in range(1, m_ + 1):
            inv[i, j] = f32(-1) / f32(f32(f32(lx[i, j] + lx[i + 1, j]) + ly[i, j]) + ly[i, j + 1])
    return r, inv, lx, ly, n_, m_


# (CB, rows, ring rows of the shipped / the diagnostic instantiation): TMC<CB>::NR in bdim_v2_tmarch.h
@pytest.mark.parametrize("CB,n_,NR", [(2, 64, 64), (2, 40, 24), (4, 56, 32), (8, 40, 28), (8, 40, 24)])
@pytest.mark.parametrize("early", [True, False])
def test_marching_down_the_rows_reproduces_the_sequential_sweeps(CB, n_, NR, early):
    r, inv, lx, ly, n_, m_ = _case(CB, n_)
    d0 = reference(r, inv, lx, ly, n_, m_)
    d1 = schedule(r, inv, lx, ly, n_, m_, CB, NR, early)
    assert np.array_equal(d0[1:-1, 1:-1].view(np.uint32), d1[1:-1, 1:-1].view(np.uint32))


def test_a_ring_one_diagonal_too_short_is_caught():
    """the model has teeth: with fewer ring rows than 2 ND + KX + 2 the late fill runs into a wait that points forwards"""
    r, inv, lx, ly, n_, m_ = _case(2, 40)
    with pytest.raises(AssertionError):
        schedule(r, inv, lx, ly, n_, m_, 2, 12, False)


def test_sweeps_two_rows_apart_are_caught():
    """... and so has the three-row distance between consecutive sweeps (TMC::LG): at two rows sweep k + 1 fetches the first
    column of the next block in the very step in which sweep k writes it -- a step's reads come before its writes, so it
    sees the value of the sweep before: other bits"""
    r, inv, lx, ly, n_, m_ = _case(2, 40)
    d0 = reference(r, inv, lx, ly, n_, m_)
    try:
        d1 = schedule(r, inv, lx, ly, n_, m_, 2, 64, True, LG=2)
    except AssertionError:
        return
    assert not np.array_equal(d0[1:-1, 1:-1].view(np.uint32), d1[1:-1, 1:-1].view(np.uint32))
