"""Boundary tests (CPU): Java number formatting, the draw()-loop bookkeeping and the XML-RPC
call sequence of the driver against a stand-in for server/server.py.  The env behind the driver
is the CPU oracle here (tests may use it); on the GPU the same driver runs on AFCEnv
(tests/test_gpu_driver.py)."""
import re

import numpy as np
import pytest

from oracle.draw_loop import OracleDrawLoop
from rlfluidcontrol_amd.driver_rpc import LilypadRPCDriver
from rlfluidcontrol_amd.javafmt import java_float_to_string, java_parse_float
from tests.standin_server import StandInServer


def test_java_float_formatting_matches_reference_checkpoint(init_bdim):
    # the first lines of the reference's init.bdim were written by Java's float->String
    head = [str(x) for x in init_bdim["head_text"]]
    assert java_float_to_string(init_bdim["t"]) == head[0] == "2400.2554"
    assert java_float_to_string(init_bdim["dt"]) == head[1] == "0.17999999"
    u, v, p = init_bdim["u"].ravel(), init_bdim["v"].ravel(), init_bdim["p"].ravel()
    for k, line in enumerate(head[2:]):
        assert "%s, %s, %s" % tuple(java_float_to_string(a[k]) for a in (u, v, p)) == line
    n_m = u.size
    tail = [str(x) for x in init_bdim["tail_text"]]
    for q, line in enumerate(tail):
        k = n_m - len(tail) + q
        assert "%s, %s, %s" % tuple(java_float_to_string(a[k]) for a in (u, v, p)) == line
    for s in ["8.5E-4", "1.0E-5", "-0.456", "1.0", "1e-05", "0.123"]:
        assert java_parse_float(java_float_to_string(java_parse_float(s))) == java_parse_float(s)
    assert java_float_to_string(8.5e-4) == "8.5E-4" and java_float_to_string(1e7) == "1.0E7"


def test_episode_arithmetic_float32():
    """for every res, 6 667 sim-steps per episode, first t>1 at step 134, first
    request after step 149, 408 requests -- pure float32 time bookkeeping, no CFD needed."""
    f32 = np.float32
    for res in (8, 16, 24, 32):
        dt = f32(f32(0.0075) * f32(res))
        t, steps, call, reqs, first_gt1, first_req = f32(0), 0, 16, 0, None, None
        while t < f32(50):
            steps += 1
            t = f32(t + f32(dt / f32(res)))
            if t > f32(1):
                first_gt1 = first_gt1 or steps
                call -= 1
                if call <= 0:
                    call = 16
                    reqs += 1
                    first_req = first_req or steps
        assert (steps, first_gt1, first_req, reqs) == (6667, 134, 149, 408), res


@pytest.fixture()
def server():
    s = StandInServer().start()
    yield s
    s.stop()


def test_rpc_transcript_against_standin_server(server):
    env = OracleDrawLoop(resolution=8, Re=100, init_time=0.05, end_time=0.62)
    drv = LilypadRPCDriver(env, url="http://127.0.0.1:%d" % server.port, train_steps=1000)
    drv.run(episodes=1)
    names = [c[0] for c in server.calls]
    # clientCFD.pde:21-25 init(-1); :106-110 start_episode(-1); requests; :73-77 train(1000), save; new sim
    assert names[0] == "init" and server.calls[0][1] == -1
    assert names[1] == "start_episode" and server.calls[1][1] == -1
    reqs = [c for c in server.calls if c[0] == "request_stochastic_action"]
    # float32 bookkeeping: dt/res = 0.0075, t>0.05 from step 7, windows of 16 until t>=0.62 (83 steps)
    assert len(reqs) == 4
    for c in reqs:
        assert re.fullmatch(r"-?\d+\.\d+(E-?\d+)?_-?\d+\.\d+(E-?\d+)?", c[1]), c[1]
    i_train = names.index("train")
    assert server.calls[i_train][1] == 1000 and names[i_train + 1] == "save" and names[i_train + 2] == "start_episode"
    assert len(server.rewards) == len(reqs) - 1  # record_length - 1 transitions (server.py:157)
    # the action strings were parsed back to float32 and applied: the driver's transcript holds them
    sent = [t for t in drv.transcript if t[0] == "request_stochastic_action"]
    assert all(isinstance(t[2], str) and "_" in t[2] for t in sent)


def test_failed_rpc_gives_zero_action():
    env = OracleDrawLoop(resolution=8, Re=100, init_time=0.0, end_time=0.2)
    drv = LilypadRPCDriver(env, url="http://127.0.0.1:9")  # nothing listens there
    xi = drv.call_action(0.1, 1.2)  # clientCFD.pde:118-132: exception -> XI stays (0,0)
    assert np.all(xi == 0)


def test_carry_quirk_emulation():
    """clientCFD.pde:41-47: Cd/Cl are not cleared after a request, so the next window starts from
    the previous normalised value; off by default, reproducible on demand."""
    a = OracleDrawLoop(resolution=8, Re=100, init_time=0.0, end_time=1.0, carry_quirk=False)
    b = OracleDrawLoop(resolution=8, Re=100, init_time=0.0, end_time=1.0, carry_quirk=True)
    z = np.zeros((1, 2), np.float32)
    oa1, _, _, n1 = a.agent_step(z)
    ob1, _, _, _ = b.agent_step(z)
    assert n1[0] == 16 and np.array_equal(oa1, ob1)
    oa2, _, _, _ = a.agent_step(z)
    ob2, _, _, _ = b.agent_step(z)
    # second window of b carries ob1/16*2/res on top
    carry = ob1[0] / np.float32(16) * np.float32(2) / np.float32(8)
    assert np.allclose(ob2[0] - oa2[0], carry, rtol=2e-3, atol=1e-6)
