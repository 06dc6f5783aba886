"""GPU: the batched agent step (draw() loop on the device) vs the CPU draw-loop oracle, episode
bookkeeping, masked reset, checkpoint round trip."""
import os

import numpy as np
import pytest

from oracle.draw_loop import OracleDrawLoop

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def envmod():
    from rlfluidcontrol_amd import build, env
    build.build()
    return env


@pytest.mark.parametrize("quirk", [0, 1])
def test_agent_step_matches_draw_loop(envmod, quirk):
    """parity mode: obs/reward/step counts of 5 agent steps are bit-identical to the oracle's
    float32 restatement of clientCFD.pde:35-56 (incl. the optional Cd/Cl carry quirk)"""
    res, Re = 8, 100
    g = envmod.AFCEnv(num_envs=2, resolution=res, reynolds=Re, exact_sums=1, init_time=0.2, end_time=50.0,
                      carry_quirk=quirk)
    o = OracleDrawLoop(resolution=res, Re=Re, init_time=0.2, end_time=50.0, carry_quirk=bool(quirk))
    rng = np.random.default_rng(7)
    a = np.zeros((1, 2), np.float32)
    for k in range(5):
        og, rg, dg, ng = g.agent_step(np.tile(a, (2, 1)))
        oo, ro, do, no = o.agent_step(a)
        assert ng[0] == no[0] and ng[1] == no[0]
        assert np.array_equal(og[0], oo[0]) and np.array_equal(og[1], oo[0]), (k, og, oo)
        assert np.array_equal(rg[:1], ro) and dg[0] == do[0]
        a = rng.uniform(-1, 1, (1, 2)).astype(np.float32)
    assert ng[0] == 16
    g.close()


def test_episode_end_and_masked_reset(envmod):
    g = envmod.AFCEnv(num_envs=3, resolution=8, reynolds=100, init_time=0.0, end_time=0.3)
    z = np.zeros((3, 2), np.float32)
    obs, rew, done, ns = g.agent_step(z)
    assert list(ns) == [16, 16, 16] and not done.any()
    obs, rew, done, ns = g.agent_step(z)
    assert not done.any()
    obs3, rew, done, ns = g.agent_step(z)  # t reaches 0.3 before the third window completes
    f32 = np.float32
    t, n_end = f32(0), 0
    while t < f32(0.3):  # float32 time bookkeeping of AFCCylinder.pde:56
        t = f32(t + f32(f32(f32(0.0075) * f32(8)) / f32(8)))
        n_end += 1
    assert 32 < n_end < 48 and list(ns) == [n_end - 32] * 3 and done.all()
    assert np.array_equal(obs3, obs)  # obs keeps the last completed window
    t_before = g.t.copy()
    g.reset(np.array([1, 0, 1], np.uint8))
    t_after = g.t
    assert t_after[0] == 0 and t_after[2] == 0 and t_after[1] == t_before[1]
    obs, rew, done, ns = g.agent_step(z)
    assert list(ns) == [16, 0, 16] and list(done) == [0, 1, 0]
    u0, v0, p0 = g.get_state(0)
    u2, v2, p2 = g.get_state(2)
    assert np.array_equal(u0, u2) and np.array_equal(p0, p2)
    g.close()


def test_bdim_roundtrip(envmod, init_bdim, tmp_path):
    g = envmod.AFCEnv(num_envs=2, resolution=24, reynolds=500)
    g.set_reset_state(init_bdim["u"], init_bdim["v"], init_bdim["p"])
    g.update2(np.array([[1.0, -1.0], [0.0, 0.0]], np.float32))
    path = os.path.join(tmp_path, "s.bdim")
    g.write(path, env_idx=0)
    u, v, p = g.get_state(0)
    h = envmod.AFCEnv(num_envs=1, resolution=24, reynolds=500)
    h.resume(path)
    u2, v2, p2 = h.get_state(0)
    assert np.array_equal(u, u2) and np.array_equal(v, v2) and np.array_equal(p, p2)
    g.close()
    h.close()


def test_developed_wake_fixture_is_a_fixed_point_class(envmod):
    """the committed res-16 wake: stepping it stays bounded and sheds (Cl changes sign over a period)"""
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "wake_res16_re100.npz"))
    g = envmod.AFCEnv(num_envs=1, resolution=16, reynolds=100, init_time=0.0, end_time=1e9)
    g.set_reset_state(z["u"], z["v"], z["p"])
    cls = []
    for _ in range(60):
        obs, _, _, _ = g.agent_step(np.zeros((1, 2), np.float32))
        cls.append(float(obs[0, 0]))
    assert g.diverged()[0] == 0 and max(np.abs(cls)) < 2.0
    assert min(cls) < -0.05 and max(cls) > 0.05
    assert 1.0 < float(obs[0, 1]) < 2.5
    g.close()


def test_td3_in_the_loop_on_device(envmod):
    """cfg5 plumbing: E envs + the TD3 learner on one GPU, two short episodes; records, transitions and
    training all stay on the device; rewards are the reference's formula applied to the recorded obs."""
    import torch

    from rlfluidcontrol_amd import td3, train

    learner, hist = train.train(num_envs=8, resolution=8, reynolds=100, episodes=2, train_steps=4, seed=1, batch_size=16,
                                env_kwargs=dict(init_time=0.05, end_time=0.62), log=lambda s: None)
    assert [h["requests"] for h in hist] == [4, 4]  # same float32 bookkeeping as the single-env driver test
    ag = learner.agent
    assert ag.replay_buffer.size == 2 * 3 * 8 and ag.total_step_count == 8 and ag.episode_count == 2
    assert ag.replay_buffer.obs1_buf.is_cuda
    n = ag.replay_buffer.size
    s2, a, r = ag.replay_buffer.obs2_buf[:n], ag.replay_buffer.acts_buf[:n], ag.replay_buffer.rews_buf[:n, 0]
    assert torch.allclose(r, td3.reward_func(s2, a), atol=1e-6)
    assert float(a.abs().max()) <= 1.0 and torch.isfinite(s2).all()
    # envs received different exploration noise, so their wakes (observations) differ
    assert float(ag.replay_buffer.obs2_buf[n - 8:n, 1].std()) > 0


def test_savescalar_log_and_vorticity_picture_from_the_env(envmod, tmp_path):
    """the sketch's per-frame log line and its picture, produced from the GPU env and checked against the oracle"""
    from oracle.oracle_py import Oracle
    from rlfluidcontrol_amd import logs

    g = envmod.AFCEnv(num_envs=2, resolution=8, reynolds=100, exact_sums=1)
    o = Oracle(resolution=8, Re=100)
    lg = logs.SaveScalarLog(str(tmp_path / "1.txt"))
    logs.record_run(g, lg, 3, xi=(0.5, -0.25))
    lg.finish()
    rows = [l.split() for l in (tmp_path / "1.txt").read_text().split("\n")[4:] if l]
    assert len(rows) == 3
    for k, toks in enumerate(rows):
        fo, _, _ = o.update2(2.5, -1.25)  # xi_m = 5 * xi (clientCFD.pde:53-54)
        assert len(toks) == 37
        assert np.float32(toks[0]) == np.float32(o.t)
        assert np.float32(toks[1]) == fo[0] and np.float32(toks[2]) == fo[1]
        assert (toks[3], toks[4]) == ("0.5", "-0.25")
        assert np.array_equal(np.array(toks[5:], np.float32), o.probes(32))
    h, w = logs.vorticity_png(str(tmp_path / "result.png"), g.vorticity(0), zoom=3)
    assert (h, w) == (3 * g.m, 3 * g.n) and (tmp_path / "result.png").stat().st_size > 200
    g.close()
