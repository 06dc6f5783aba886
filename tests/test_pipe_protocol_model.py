"""The hand-over protocol of the pipelined up-leg (mg_up_pipe, rlfluidcontrol_amd/csrc/bdim_v2_body.h) as a discrete-event
model: one marching wave and three helper waves advance under a RANDOM scheduler (with spells in which some of the waves
all but stand still); every ring cell (ring slot x column)
remembers which row it holds and how many sweeps the march has completed on it, and every access asserts that it sees
what the sequential algorithm (fill -> four ordered sweeps -> exit, strip by strip: mg_up_smooth) would give it:

  * a march step of sweep k updates a cell that holds its own row after exactly k sweeps, takes d[i][j+1] and d[i+1][j]
    after exactly k sweeps (the previous sweep's values) and, for the first row of the strip, d[i-1][j] after k+1 (the
    previous strip's result); its operand fetches run ahead of it, so the rows it walks must be in the ring up to 13
    columns ahead of its cell (never a row the next strip's fill has replaced, never a row not filled yet);
  * an exit reads rows g-1, g, g+1 of its column group (one column to either side in row g) after all four sweeps and
    before the next strip's fill replaces them; every cell leaves exactly once;
  * a fill may overwrite a ring slot only when nobody will read the row it replaces any more (a later read of that
    row finds the wrong row in the slot and reports it).

The waits are the kernel's: progress >= 16 q + 44 for the exit of column group q; the exits of groups q-2, q-1 and q before
the fill of group q-1; the next strip's groups 0 .. 4 at the head of a march and the rest at step 64.  The model has
teeth: with a threshold of 16 q + 30, without the wait for the exit of group q-2, or with the march's waits weakened, it
reports hazards.  CPU only -- this checks the PROTOCOL; the GPU suite checks the KERNEL (bit-exact against the oracle and
against the barrier form)."""
import random

import pytest

H, NR, LAG, AHEAD = 16, 24, 4, 13  # rows per strip, ring rows, steps between consecutive sweeps, columns the fetches run ahead


class Hazard(Exception):
    pass


class Model:
    def __init__(self, n_, m_, rng, exit_need=44, wait_q2=True, mid_wait=True, head_groups=5):
        self.n_, self.m_, self.Q = n_, m_, m_ // 16
        self.rng = rng
        self.exit_need, self.wait_q2, self.mid_wait, self.head_groups = exit_need, wait_q2, mid_wait, head_groups
        self.steps = (m_ + (H - 1) + 3 * LAG + 3) & ~3
        self.strips = [I0 for I0 in range(1, n_ + 4, H) if I0 - 3 <= n_]
        self.row = [[0] * (m_ + 2) for _ in range(NR)]    # which row a ring cell holds (0: none)
        self.done = [[0] * (m_ + 2) for _ in range(NR)]   # sweeps completed on it
        for g in range(1, min(H + 1, n_) + 1):             # prologue: rows 1 .. H+1
            self.row[g % NR] = [g] * (m_ + 2)
        self.progress = 0            # (strip << 10) + steps completed, as the kernel publishes it (every 8 steps)
        self.xf = [0] * (self.Q + 1)
        self.ff = [0] * self.Q
        self.left = [[False] * (m_ + 2) for _ in range(n_ + 2)]

    def present(self, i, j, who):
        if 1 <= i <= self.n_ and 1 <= j <= self.m_ and self.row[i % NR][j] != i:
            raise Hazard("%s touches row %d col %d but the slot holds row %d" % (who, i, j, self.row[i % NR][j]))

    def value(self, i, j, sweeps, who):
        if 1 <= i <= self.n_ and 1 <= j <= self.m_:
            self.present(i, j, who)
            if self.done[i % NR][j] != sweeps:
                raise Hazard("%s reads row %d col %d after %d sweeps, expects %d" % (who, i, j, self.done[i % NR][j], sweeps))

    # ---- the marching wave (a generator: yields wherever another wave may run) ---------------------------------------
    def march(self):
        n_, m_ = self.n_, self.m_
        for s, I0 in enumerate(self.strips):
            head = min(self.Q, self.head_groups)
            while any(self.ff[q] < s for q in range(head)):
                yield
            for tau in range(self.steps):
                if tau == 64 and self.mid_wait:
                    while any(self.ff[q] < s for q in range(head, self.Q)):
                        yield
                who = "march(strip %d, step %d)" % (s, tau)
                for k in range(4):          # sweep k follows sweep k-1 at LAG steps: its lanes read what that sweep has stored
                    for c in range(H):
                        i, j = I0 - k + c, tau + 1 - c - LAG * k
                        if i < 1 or i > n_:
                            continue
                        for a in range(1, AHEAD + 1):   # the rows it walks are in the ring ahead of it (operand fetches)
                            self.present(i, j + a, who)
                            self.present(i + 1, j + a - 1, who)
                        if 1 <= j <= m_:
                            self.value(i, j, k, who)            # its own cell: sweep k-1 done, sweep k not yet
                            self.value(i, j + 1, k, who)        # d[i][j+1]: the previous sweep's value
                            self.value(i + 1, j, k, who)        # d[i+1][j]: the previous sweep's value
                            if c == 0:
                                self.value(i - 1, j, k + 1, who)  # d[i-1][j] of THIS sweep, finished by the previous strip
                for k in range(4):          # ... and all lanes store at once
                    for c in range(H):
                        i, j = I0 - k + c, tau + 1 - c - LAG * k
                        if 1 <= i <= n_ and 1 <= j <= m_:
                            self.done[i % NR][j] = k + 1
                if (tau + 1) % 8 == 0:
                    self.progress = (s << 10) + tau + 1
                    yield
            self.progress = (s + 1) << 10
            yield

    # ---- a helper wave: tasks q = hw, hw + 3, ... of every strip: exit of group q, then fill of group q-1 ------------------
    def helper(self, hw):
        n_, m_ = self.n_, self.m_
        for s, I0 in enumerate(self.strips):
            e0, e1 = max(1, I0 - 4), (n_ if I0 + H - 4 >= n_ else I0 + H - 5)
            last = I0 + H - 3 > n_
            for q in range(hw, self.Q + 1, 3):
                if q < self.Q:
                    while self.progress < (s << 10) + 16 * q + self.exit_need:
                        yield
                    who = "exit(strip %d, group %d)" % (s, q)
                    for g in range(e0, e1 + 1):
                        gw, ge = max(g - 1, 1), min(g + 1, n_)
                        for col in range(16 * q + 1, 16 * q + 17):
                            self.value(gw, col, 4, who)
                            self.value(g, col, 4, who)
                            self.value(ge, col, 4, who)
                            if self.left[g][col]:
                                raise Hazard("row %d col %d leaves twice" % (g, col))
                            self.left[g][col] = True
                        self.value(g, 16 * q, 4, who)
                        self.value(g, 16 * q + 17, 4, who)
                    yield
                    self.xf[q] = s + 1
                    yield
                if q >= 1 and not last:
                    while self.xf[q - 1] < s + 1 or (self.wait_q2 and q >= 2 and self.xf[q - 2] < s + 1):
                        yield
                    for gf in range(I0 + H + 1, I0 + 2 * H + 1):
                        if gf <= n_:
                            for col in range(16 * (q - 1) + 1, 16 * q + 1):
                                self.row[gf % NR][col] = gf
                                self.done[gf % NR][col] = 0
                    yield
                    self.ff[q - 1] = s + 1
                    yield

    def run(self):
        waves = [self.march(), self.helper(0), self.helper(1), self.helper(2)]
        alive = [0, 1, 2, 3]
        spins = 0
        weight = [1.0] * 4
        while alive:
            if spins % 500 == 0:  # an adversarial scheduler: every now and then some waves all but stand still
                weight = [self.rng.choice((1.0, 1.0, 0.2, 0.01)) for _ in range(4)]
            w = self.rng.choices(alive, [weight[a] for a in alive])[0]
            try:
                next(waves[w])
            except StopIteration:
                alive.remove(w)
            spins += 1
            if spins > 6_000_000:
                raise Hazard("no progress (deadlock?)")
        for g in range(1, self.n_ + 1):
            if not all(self.left[g][1:self.m_ + 1]):
                raise Hazard("row %d never left" % g)


@pytest.mark.parametrize("n_,m_", [(32, 32), (64, 32), (128, 64), (64, 128), (256, 128)])
def test_protocol_has_no_hazard_under_random_schedules(n_, m_):
    for seed in range(2 if n_ * m_ >= 8192 else 5):
        Model(n_, m_, random.Random(1000 * n_ + seed)).run()


def _hazard(**kw):
    for seed in range(12):
        try:
            Model(64, 128, random.Random(seed), **kw).run()
        except Hazard:
            return True
    return False


def test_the_model_has_teeth():
    """weakened protocols are caught: an exit that starts 14 steps early reads rows sweep 3 has not finished; a fill that
    does not wait for the exit two groups back replaces a column that exit still reads; a march that waits for fewer
    column groups walks rows the fill has not delivered"""
    assert _hazard(exit_need=30)
    assert _hazard(wait_q2=False)
    assert _hazard(mid_wait=False, head_groups=2)
