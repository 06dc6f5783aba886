"""The C library's Java float text (afc_format_float_java: what BDIM.write's `""+u.x.a[i][j]` and
clientCFD.pde:120's String.valueOf print) against the reference's own checkpoint lines and against the
Python formatter the RPC driver uses.  Host code only: runs without a GPU."""
import numpy as np

from rlfluidcontrol_amd import build, env
from rlfluidcontrol_amd.javafmt import java_float_to_string


def test_c_formatter_reproduces_reference_checkpoint_text(init_bdim):
    build.build()
    n, m = init_bdim["u"].shape
    # header: t, dt (BDIM.pde:229-230)
    assert env.java_float_text(init_bdim["t"]) == str(init_bdim["head_text"][0])
    assert env.java_float_text(init_bdim["dt"]) == str(init_bdim["head_text"][1])
    u, v, p = (init_bdim[k].reshape(-1) for k in ("u", "v", "p"))
    head = [str(s) for s in init_bdim["head_text"][2:]]
    for k, want in enumerate(head):  # first cells, in file order (i outer, j inner)
        got = ", ".join(env.java_float_text(a[k]) for a in (u, v, p))
        assert got == want, (k, got, want)
    tail = [str(s) for s in init_bdim["tail_text"]]
    for q, want in enumerate(tail):
        k = n * m - len(tail) + q
        got = ", ".join(env.java_float_text(a[k]) for a in (u, v, p))
        assert got == want, (k, got, want)


def test_c_formatter_equals_python_formatter_on_many_floats():
    build.build()
    rng = np.random.default_rng(5)
    xs = np.concatenate([
        rng.standard_normal(4000).astype(np.float32),
        (rng.standard_normal(6000) * 10.0 ** rng.integers(-12, 12, 6000)).astype(np.float32),
        rng.integers(0, 2 ** 32, 6000, dtype=np.uint64).astype(np.uint32).view(np.float32),
        np.array([0.0, -0.0, 1.0, 1e7, 9999999.0, 1e-3, 9.999e-4, 123456.7, 1e10, 3.4028235e38, 1.4e-45,
                  1.17549435e-38, 0.1, 0.3, 2e-3, 8.5e-4, -2678052.75, 5910.84375, 0.17999999], np.float32)])
    for x in xs:
        if np.isfinite(x):
            assert env.java_float_text(x) == java_float_to_string(x), repr(float(x))
    assert env.java_float_text(np.float32(8.5e-4)) == "8.5E-4"  # the exponent form Float.parseFloat / float() read back
    assert env.java_float_text(np.float32(1e7)) == "1.0E7" and env.java_float_text(np.float32(0.001)) == "0.001"
    assert env.java_float_text(np.inf) == "Infinity" and env.java_float_text(np.nan) == "NaN"
