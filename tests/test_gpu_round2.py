"""GPU tests added in round 2: the CFL read-out, the Java-text checkpoint writer and reference-format reader,
the XML-RPC boundary over the HIP env (BASELINE cfg1), the window-on-last-step protocol edge, the full-size
cfg4 / cfg5 workloads, the default-mode tolerance at res 32, and the 2-rank RCCL gather (when 2 GPUs exist)."""
import os
import re
import socket

import numpy as np
import pytest

from oracle.draw_loop import OracleDrawLoop
from tests.standin_server import StandInServer

pytestmark = pytest.mark.gpu
f32 = np.float32


@pytest.fixture(scope="module")
def envmod():
    from rlfluidcontrol_amd import build, env
    build.build()
    return env


def test_cfl_matches_oracle(envmod, oracle_mod):
    """BDIM.checkCFL / VectorField.CFL (VectorField.pde:225-235) of the live state, bit for bit"""
    g = envmod.AFCEnv(num_envs=3, resolution=8, reynolds=100, exact_sums=1)
    o = oracle_mod.Oracle(resolution=8, Re=100)
    assert np.array_equal(g.cfl(), np.full(3, o.cfl(), np.float32))  # free stream: 1/(1 + 3 nu)
    xi = np.array([[4.0, -3.0]] * 3, np.float32)
    for _ in range(3):
        g.update2(xi)
        o.update2(4.0, -3.0)
        c = g.cfl()
        assert c[0] == np.float32(o.cfl()) and c[0] == c[1] == c[2]
    assert 0.2 < c[0] < 0.8  # |u|+|v| of a few near the spinning cylinders
    g.close()


def test_bdim_writer_is_java_text_and_reference_format_reads_back(envmod, init_bdim, tmp_path):
    """BDIM.write/resume (BDIM.pde:226-251): our writer reproduces the reference checkpoint's own lines for the
    same values, and a file in the reference's number format (incl. E-notation) loads to the same bits."""
    g = envmod.AFCEnv(num_envs=1, resolution=24, reynolds=500)
    g.set_reset_state(init_bdim["u"], init_bdim["v"], init_bdim["p"])
    path = os.path.join(tmp_path, "w.bdim")
    g.write(path)
    lines = open(path).read().split("\n")
    n, m = init_bdim["u"].shape
    assert len(lines) == 2 + n * m + 1 and lines[-1] == ""
    assert lines[1] == str(init_bdim["head_text"][1])  # dt = 0.17999999
    head = [str(s) for s in init_bdim["head_text"][2:]]
    assert lines[2:2 + len(head)] == head
    tail = [str(s) for s in init_bdim["tail_text"]]
    assert lines[2 + n * m - len(tail):2 + n * m] == tail
    assert all(re.fullmatch(r"(-?\d+\.\d+(E-?\d+)?)(, -?\d+\.\d+(E-?\d+)?){2}", l) for l in lines[2:2 + n * m:997])
    # a state with tiny values prints E-notation ("8.5E-4" style); the reader must take it
    rng = np.random.default_rng(3)
    u = (init_bdim["u"] * 1e-4).astype(np.float32)
    v = (rng.standard_normal((n, m)) * 1e-6).astype(np.float32)
    p = (init_bdim["p"] * 3e7).astype(np.float32)
    g.set_reset_state(u, v, p)
    g.write(path)
    txt = open(path).read()
    assert "E-" in txt and re.search(r"\dE\d", txt)
    h = envmod.AFCEnv(num_envs=2, resolution=24, reynolds=500)
    h.resume(path)
    for e in range(2):
        u2, v2, p2 = h.get_state(e)
        assert np.array_equal(u2, u) and np.array_equal(v2, v) and np.array_equal(p2, p)
    g.close()
    h.close()


def test_mg_histogram_counts_every_solve(envmod):
    g = envmod.AFCEnv(num_envs=4, resolution=8, reynolds=100)
    assert g.mg_histogram().sum() == 0
    xi = np.zeros((4, 2), np.float32)
    for _ in range(5):
        g.update2(xi)
    h = g.mg_histogram()
    assert h.sum() == 4 * 5 * 2 and h[0] > 0  # two pressure solves per env per sim-step
    it, _ = g.solver_stats()
    assert h[int(it.max()) - 1] > 0
    g.agent_step(np.zeros((4, 2), np.float32))
    h2 = g.mg_histogram(clear=True)
    assert h2.sum() > h.sum() and g.mg_histogram().sum() == 0
    g.close()


def test_rpc_boundary_over_the_hip_env_cfg1(envmod):
    """BASELINE cfg1 plumbing: our counterpart of clientCFD.pde drives ONE HIP env (128x64, Re 100) against the
    stand-in for server.py over real XML-RPC; in parity mode the whole transcript (every "Cl_Cd" string sent and
    every action received) is identical to the run with the CPU draw-loop oracle behind the same driver."""
    from rlfluidcontrol_amd.driver_rpc import LilypadRPCDriver

    def run(env):
        srv = StandInServer().start()
        try:
            drv = LilypadRPCDriver(env, url="http://127.0.0.1:%d" % srv.port, train_steps=1000)
            drv.run(episodes=1)
            return drv.transcript, list(srv.calls), list(srv.rewards)
        finally:
            srv.stop()

    kw = dict(init_time=0.05, end_time=0.62)
    g = LilypadRPCDriver.reference_env(resolution=8, reynolds=100, exact_sums=1, **kw)
    assert g.cfg.carry_quirk == 1
    tg, cg, rg = run(g)
    to, co, ro = run(OracleDrawLoop(resolution=8, Re=100, carry_quirk=True, **kw))
    assert tg == to and cg == co and rg == ro
    names = [c[0] for c in cg]
    assert names[:2] == ["init", "start_episode"] and names.count("request_stochastic_action") == 4
    assert names[names.index("train") + 1] == "save"
    g.close()


def test_window_completing_on_the_last_step_is_still_sent(envmod):
    """clientCFD.pde:36-56: draw() asks for the action inside the step and only ends the episode on the next
    frame, so a window that completes on the step that takes t to Time is sent; one cut short is not."""
    from rlfluidcontrol_amd.driver_rpc import LilypadRPCDriver
    t = f32(0)
    per = f32(f32(f32(0.0075) * f32(8)) / f32(8))
    for _ in range(32):
        t = f32(t + per)
    for end, want_reqs in ((float(t), 2), (float(t) + 0.004, 2), (float(t) - 0.004, 2), (float(t) - 0.01, 1)):
        kw = dict(init_time=-1.0, end_time=end)
        res = []
        for env in (LilypadRPCDriver.reference_env(resolution=8, reynolds=100, exact_sums=1, **kw),
                    OracleDrawLoop(resolution=8, Re=100, carry_quirk=True, **kw)):
            srv = StandInServer().start()
            try:
                drv = LilypadRPCDriver(env, url="http://127.0.0.1:%d" % srv.port)
                drv.setup()
                n = drv.run_episode()
                res.append((n, [c for c in srv.calls if c[0].startswith("request")]))
            finally:
                srv.stop()
            if hasattr(env, "close"):
                env.close()
        assert res[0] == res[1] and res[0][0] == want_reqs, (end, res)


def test_default_mode_tolerance_at_res32(envmod, oracle_mod):
    """the mode bench.py times (tree sums for Field.sum / Field.inner) on the 512x256 Re 400 grid, 16 steps with
    actuation, against the oracle: same stated tolerance as on the smaller grids (tests/test_gpu_parity.py)"""
    res, Re = 32, 400
    g = envmod.AFCEnv(num_envs=2, resolution=res, reynolds=Re)
    o = oracle_mod.Oracle(resolution=res, Re=Re)
    rng = np.random.default_rng(11)
    mx = 0.0
    for k in range(16):
        a = (0.0, 0.0) if k < 2 else tuple(5 * rng.uniform(-1, 1, 2))
        fg = g.update2(np.tile(np.array([a], np.float32), (2, 1)))
        fo, ito, _ = o.update2(float(f32(a[0])), float(f32(a[1])))
        assert np.array_equal(fg[0], fg[1])
        sc = max(1.0, float(np.abs(fo).max()))
        assert np.abs(fg[0].astype(np.float64) - fo).max() <= 5e-5 * sc * res / 2, (k, fg[0], fo)  # forces are O(res)
        mx = max(mx, float(np.abs(fg[0].astype(np.float64) - fo).max()) / sc)
    ug, vg, pg = g.get_state(0)
    uo, vo, po = o.get_state()
    assert np.abs(ug.astype(np.float64) - uo).max() <= 5e-6 and np.abs(vg.astype(np.float64) - vo).max() <= 5e-6
    assert np.abs(pg.astype(np.float64) - po).max() <= 5e-4 * max(1.0, float(np.abs(po).max()))
    itg, _ = g.solver_stats()
    assert tuple(itg[0]) == tuple(ito)
    g.close()


def test_cfg4_full_size_1024_envs_512x256(envmod):
    """BASELINE cfg4 at its own size: 1024 envs on the 512x256 Re 400 grid (level 1 streams, 8 MG levels).  All envs
    get the same inputs: they must stay bit-identical to each other and to a single env run alone, with no
    divergence, one V-cycle per solve after the start-up."""
    E, res, Re = 1024, 32, 400
    g = envmod.AFCEnv(num_envs=E, resolution=res, reynolds=Re)
    one = envmod.AFCEnv(num_envs=1, resolution=res, reynolds=Re)
    acts = [(0.0, 0.0), (2.5, -4.0), (2.5, -4.0), (-1.0, 3.0)]
    for a in acts:
        fg = g.update2(np.tile(np.array([a], np.float32), (E, 1)))
        f1 = one.update2(np.array([a], np.float32))
        assert np.all(fg == fg[0]) and np.array_equal(fg[0], f1[0])
    assert g.diverged().sum() == 0
    for e in (0, 255, 256, 777, 1023):
        for x, y in zip(g.get_state(e), one.get_state(0)):
            assert np.array_equal(x, y)
    it, rs = g.solver_stats()
    assert it.max() <= 2 and np.all(np.isfinite(rs))
    c = g.cfl()
    assert np.all(c == c[0]) and c[0] == one.cfl()[0]
    g.close()
    one.close()


def test_native_full_size_1024_envs_384x192(envmod):
    """the reference's own case at the batch size of the headline: 1024 envs on the 384x192 Re 500 grid (clientCFD.pde:5-14;
    96-column tiles, 12-row strips on the finest level, an odd-width coarsest level), started from the reference-held
    init.bdim.  All envs get the same inputs: bit-identical to each other and to a single env run alone, no divergence,
    one V-cycle per solve from a developed state."""
    from tests.conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, "init_bdim_res24.npz"))
    E, res, Re = 1024, 24, 500
    g = envmod.AFCEnv(num_envs=E, resolution=res, reynolds=Re)
    one = envmod.AFCEnv(num_envs=1, resolution=res, reynolds=Re)
    for h in (g, one):
        h.set_reset_state(z["u"], z["v"], z["p"])
        h.reset()
    acts = [(0.0, 0.0), (2.5, -4.0), (2.5, -4.0), (-1.0, 3.0)]
    for a in acts:
        fg = g.update2(np.tile(np.array([a], np.float32), (E, 1)))
        f1 = one.update2(np.array([a], np.float32))
        assert np.all(fg == fg[0]) and np.array_equal(fg[0], f1[0])
    assert g.diverged().sum() == 0
    for e in (0, 255, 256, 777, 1023):
        for x, y in zip(g.get_state(e), one.get_state(0)):
            assert np.array_equal(x, y)
    it, rs = g.solver_stats()
    assert it.max() <= 2 and np.all(np.isfinite(rs))
    g.close()
    one.close()


def test_impl2_rejects_grids_taller_than_256_cells_and_impl0_runs_them(envmod):
    """AFCCylinder.pde:12-13 sizes the grid from yLengths*resolution without a cap; the optimised kernels are instantiated
    for up to 256 cells in y (four columns per lane).  A taller grid must be refused LOUDLY by impl 2 (never computed
    wrongly) and is served by the straightforward kernels (impl 0), which have no such limit."""
    with pytest.raises(envmod.AfcError) as ei:
        envmod.AFCEnv(num_envs=1, resolution=40, reynolds=100, impl=2)  # 640 x 320 (2^p x 5: a valid multigrid size)
    assert "256 cells in y" in str(ei.value)
    g = envmod.AFCEnv(num_envs=1, resolution=40, reynolds=100, impl=0)
    assert (g.n, g.m) == (16 * 40 + 2, 8 * 40 + 2)
    f = g.update2(np.zeros((1, 2), np.float32))
    assert np.all(np.isfinite(f))
    u, v, p = g.get_state(0)
    assert np.all(np.isfinite(u)) and np.all(np.isfinite(p))
    g.close()


def test_cfg5_td3_in_the_loop_at_1024_envs(envmod):
    """BASELINE cfg5's per-GPU workload: 1024 envs x 256x128 with the TD3 learner on the same device -- the
    episode protocol up to the 8th request (149 + 7x16 sim-steps per env), transitions assembled, 8 TD3 iterations."""
    import torch

    from rlfluidcontrol_amd import td3, train
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "wake_res16_re100.npz"))
    learner, hist = train.train(num_envs=1024, resolution=16, reynolds=100, episodes=1, train_steps=8, seed=2,
                                max_requests=8, init=z, log=lambda s: None)
    h = hist[0]
    assert h["requests"] == 8 and h["buffer"] == 7 * 1024 and h["td3_iters"] == 8 and h["world"] == 1
    ag = learner.agent
    n = ag.replay_buffer.size
    s1, s2, a, r = (ag.replay_buffer.obs1_buf[:n], ag.replay_buffer.obs2_buf[:n], ag.replay_buffer.acts_buf[:n],
                    ag.replay_buffer.rews_buf[:n, 0])
    assert torch.isfinite(s1).all() and torch.isfinite(s2).all() and torch.isfinite(r).all()
    assert torch.allclose(r, td3.reward_func(s2, a), atol=1e-6) and float(a.abs().max()) <= 1.0
    last = s2[n - 1024:n]
    assert float(last[:, 1].std()) > 1e-4          # envs decorrelate under their own exploration noise
    assert 0.5 < float(last[:, 1].mean()) < 3.0    # Cd of a Re-100 cylinder wake


def _rank_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        from rlfluidcontrol_amd import build, env
        from rlfluidcontrol_amd.vec_env import HipBackend, ShardedVecEnv
        build.build()
        E = 4
        g = env.AFCEnv(num_envs=E, resolution=8, reynolds=100, device=rank, init_time=0.0, end_time=10.0)
        ve = ShardedVecEnv(HipBackend(g, dev), E, device=dev)
        gen = torch.Generator().manual_seed(5)
        outs = []
        for _ in range(3):
            act = (torch.rand(world * E, 2, generator=gen) * 2 - 1).to(dev)  # same tensor on every rank
            obs, rew, done = ve.step(act)
            outs.append(torch.cat([obs.reshape(-1), rew, done.float()]).cpu())
        q.put((rank, ve.n_collectives, torch.stack(outs).numpy()))
        g.close()
    finally:
        dist.destroy_process_group()


PATHS = [(4, "all-waves smoother"), (15 << 16, "CU populations taking turns"), (2, "r written back on every V-cycle"),
         (1024, "advection as LDS tiles with workgroup barriers (advdif_v2)"),
         (2048, "solve head in 14-row blocks with workgroup barriers (mg_down_fused)"),
         (128, "bottom of the V-cycle level by level through global memory instead of out of LDS (mg_bottom)"),
         (256, "up-legs with two strips per round and two marching waves, r streamed from global memory (mg_up_smooth2)"),
         (16, "coarse down-legs in blocks with workgroup barriers (mg_down_fused) instead of per-wave row pipelines (down_v3c)"),
         (32, "forcing band through gathers from the full coefficient arrays instead of the compact band tables"),
         (-(1 << 31), "u.setBC() as its own pass with strided column accesses instead of folded into the row stores")]


def _run_path(envmod, dbg, exact):
    rng = np.random.default_rng(5)
    a = rng.uniform(-1, 1, (40, 2)).astype(np.float32)
    g = envmod.AFCEnv(num_envs=40, resolution=16, reynolds=100, exact_sums=exact, init_time=0.05, dbg=dbg)
    rec = []
    for _ in range(3):
        obs, rew, done, ns = g.agent_step(a)
        rec.append((obs.copy(), rew.copy(), ns.copy()))
    u, v, p = g.get_state(7)
    hist = g.mg_histogram()
    g.close()
    return rec, u.copy(), v.copy(), p.copy(), hist


@pytest.mark.parametrize("exact", [1, 0])
@pytest.mark.parametrize("dbg,what", PATHS)
def test_experiment_paths_are_bit_identical_to_the_default(envmod, dbg, what, exact):
    """the alternative execution paths kept behind `dbg` switches change WHEN and BY WHOM the work
    is done, never a bit of the result: 40 envs with different actions, agent steps through the impulsive start (whose
    first solves take several V-cycles, so the lazy-residual replay runs in the default mode), in parity mode
    (exact_sums = 1) and in the default mode.  One exception by construction: the all-waves smoother cuts the finest
    level into strips of another height, so in the default mode its partial sums of r.r and sum(p) are added in another
    order -- there the comparison is by the default mode's tolerance, iteration counts still equal."""
    r0, u0, v0, p0, h0 = _run_path(envmod, 1 << 30, exact)  # the shipped path, through the diagnostic instantiation
    r1, u1, v1, p1, h1 = _run_path(envmod, dbg, exact)
    assert h0[1:].sum() > 0  # some solves took more than one V-cycle
    assert np.array_equal(h0, h1), what
    # ... and the two-strip up-leg hands a thread other rows of the exit, so its partial sums differ the same way
    loose = (dbg in (4, 256) and exact == 0)
    for (o0, w0, n0), (o1, w1, n1) in zip(r0, r1):
        assert np.array_equal(n0, n1), what
        if loose:
            assert np.allclose(o0, o1, rtol=5e-5, atol=5e-5) and np.allclose(w0, w1, rtol=5e-5, atol=5e-5), what
        else:
            assert np.array_equal(o0.view(np.uint32), o1.view(np.uint32)), what
            assert np.array_equal(w0.view(np.uint32), w1.view(np.uint32)), what
    for x, y, tol in ((u0, u1, 5e-6), (v0, v1, 5e-6), (p0, p1, 5e-4 * max(1.0, float(np.abs(p0).max())))):
        if loose:
            assert float(np.abs(x - y).max()) <= tol, what
        else:
            assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), what


@pytest.mark.parametrize("res", [8, 16])
@pytest.mark.parametrize("exact", [1, 0])
def test_fused_tails_through_launches_of_many_sim_steps(envmod, res, exact):
    """afc_config.fuse_tail = 1 against -1 through agent steps: inside a launch every sim-step but the last leaves the tail
    of its corrector's projection and the Heun average (BDIM.pde:95-96) to the next sim-step's predictor advection
    (advdif_v3f<C, true>), the last one runs it as a pass of its own so that the state in memory is complete.  40 envs
    with different actions from a cold start (the first launch runs 130+ sim-steps up to init_time = 1 and its first
    solves take several V-cycles), then action windows of 16: every observation, reward and step count, the multigrid
    histogram and the final fields are bit-identical; and single sim-step launches (update2) in between keep working."""
    rng = np.random.default_rng(5)
    E = 40
    acts = rng.uniform(-1, 1, (5, E, 2)).astype(np.float32)
    recs = []
    for fuse in (1, -1):
        g = envmod.AFCEnv(num_envs=E, resolution=res, reynolds=100, exact_sums=exact, fuse_tail=fuse)
        assert g.fused_tail() is (fuse == 1)
        rec = []
        for k in range(5):
            obs, rew, done, ns = g.agent_step(acts[k])
            rec.append((obs.copy(), rew.copy(), ns.copy()))
            if k == 2:
                rec.append((g.update2(5 * acts[k]).copy(), rew.copy(), ns.copy()))
        st = [g.get_state(e) for e in (0, 7, E - 1)]
        recs.append((rec, st, g.mg_histogram()))
        g.close()
    (ra, sa, ha), (rb, sb, hb) = recs
    assert ra[0][2][0] > 100 and np.array_equal(ha, hb)
    for (o1, w1, n1), (o2, w2, n2) in zip(ra, rb):
        assert np.array_equal(n1, n2) and np.array_equal(o1.view(np.uint32), o2.view(np.uint32))
        assert np.array_equal(w1.view(np.uint32), w2.view(np.uint32))
    for x, y in zip(sa, sb):
        for f1, f2 in zip(x, y):
            assert np.array_equal(f1.view(np.uint32), f2.view(np.uint32))


def test_fused_tails_when_the_launch_ends_on_the_clock(envmod):
    """the launch loop of k_agent_step ends on the action window, on the step count or on t >= end_time (clientCFD.pde:36);
    a sim-step leaves its Heun tail to the next one only if it knows that one follows.  end_time inside an action window:
    the last sim-step of the episode must have completed the state itself -- same observations, done flags, step counts and
    fields as the two-pass form, and a further call (which runs no sim-step) changes nothing."""
    E = 6
    a = np.linspace(-1, 1, 2 * E).reshape(E, 2).astype(np.float32)
    out = []
    for fuse in (1, -1):
        g = envmod.AFCEnv(num_envs=E, resolution=8, reynolds=100, fuse_tail=fuse, end_time=1.2)
        rec = []
        for k in range(4):
            obs, rew, done, ns = g.agent_step(a)
            rec.append((obs.copy(), done.copy(), ns.copy()))
        out.append((rec, [g.get_state(e) for e in (0, E - 1)]))
        g.close()
    (ra, sa), (rb, sb) = out
    assert ra[-1][1].all() and ra[-1][2].max() == 0 and any(0 < r[2].max() < 16 for r in ra)  # the clock ran out inside a window
    for (o1, d1, n1), (o2, d2, n2) in zip(ra, rb):
        assert np.array_equal(o1.view(np.uint32), o2.view(np.uint32)) and np.array_equal(d1, d2) and np.array_equal(n1, n2)
    for x, y in zip(sa, sb):
        for f1, f2 in zip(x, y):
            assert np.array_equal(f1.view(np.uint32), f2.view(np.uint32))


def test_shipped_and_diagnostic_instantiations_agree_bit_for_bit(envmod):
    """afc::prod (dbg == 0) and afc::diag (any dbg != 0) are the same code with and without the diagnostics compiled in"""
    for exact in (1, 0):
        r0, u0, v0, p0, h0 = _run_path(envmod, 0, exact)
        r1, u1, v1, p1, h1 = _run_path(envmod, 1 << 30, exact)
        assert np.array_equal(h0, h1)
        for (o0, w0, n0), (o1, w1, n1) in zip(r0, r1):
            assert np.array_equal(o0.view(np.uint32), o1.view(np.uint32)) and np.array_equal(n0, n1)
        for x, y in ((u0, u1), (v0, v1), (p0, p1)):
            assert np.array_equal(x.view(np.uint32), y.view(np.uint32))


def test_developed_wake_fixture_replays_on_the_oracle(envmod, oracle_mod):
    """tests/golden/wake_res16_re100.npz (the bench's start state) was produced by the HIP path: four parity-mode steps
    from it, with and without actuation, are bit-identical on the oracle"""
    import os
    from tests.conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, "wake_res16_re100.npz"))
    g = envmod.AFCEnv(num_envs=2, resolution=16, reynolds=100, exact_sums=1)
    o = oracle_mod.Oracle(resolution=16, Re=100)
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


def test_two_gpu_gather_every_rank_sees_the_same_batch():
    """one RCCL all-gather of (obs, reward, done) per agent step; needs 2 GPUs (skipped otherwise)"""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
    assert res[0][1] == res[1][1] == 3                # n_collectives == agent steps
    assert np.array_equal(res[0][2], res[1][2])        # identical gathered (obs, reward, done) on both ranks
    obs = res[0][2][:, :16].reshape(3, 8, 2)
    assert not np.array_equal(obs[2, :4], obs[2, 4:])  # the two shards ran different actions


def test_pipelined_upleg_at_full_occupancy_is_deterministic_and_equals_the_barrier_form(envmod):
    """race detector for mg_up_pipe (waves hand rows over through LDS flags instead of workgroup barriers) under the
    load it ships for: 1024 envs, random actions, 8 agent steps -- two runs of the shipped kernels on the same inputs
    must agree bit for bit (a timing-dependent hazard would not), and in parity mode the shipped kernels must agree bit
    for bit with the barrier form (dbg bit 8, diagnostic instantiation): every observation of every env at every agent
    step, the multigrid iteration histogram and the final fields of 32 envs.  (tools/pipe_soak.py: the same over any
    number of agent steps; tests/test_pipe_protocol_model.py: the protocol itself on the CPU.)"""
    E, N = 1024, 8
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "wake_res16_re100.npz"))

    def run(dbg, exact):
        g = envmod.AFCEnv(num_envs=E, resolution=16, reynolds=100, dbg=dbg, exact_sums=exact)
        g.set_reset_state(z["u"], z["v"], z["p"])
        g.reset()
        rng = np.random.default_rng(7)
        rec = []
        for _ in range(N):
            obs, rew, done, ns = g.agent_step(rng.uniform(-1, 1, (E, 2)).astype(np.float32))
            rec.append((obs.copy(), rew.copy(), ns.copy()))
        st = [g.get_state(e) for e in range(0, E, 32)]
        h = g.mg_histogram()
        g.close()
        return rec, st, h

    def same(a, b):
        (ra, sa, ha), (rb, sb, hb) = a, b
        assert np.array_equal(ha, hb)
        for (o1, w1, n1), (o2, w2, n2) in zip(ra, rb):
            assert np.array_equal(n1, n2) and np.array_equal(o1.view(np.uint32), o2.view(np.uint32))
            assert np.array_equal(w1.view(np.uint32), w2.view(np.uint32))
        for x, y in zip(sa, sb):
            for f1, f2 in zip(x, y):
                assert np.array_equal(f1.view(np.uint32), f2.view(np.uint32))

    a1 = run(0, 0)
    same(a1, run(0, 0))
    assert a1[2][1:].sum() > 0  # some solves took a second V-cycle: the replay of the finest up-leg ran too
    same(run(0, 1), run(256, 1))


def _run_upleg(envmod, upleg, exact, res, E, N=3, init_time=0.05, start=None):
    rng = np.random.default_rng(11)
    g = envmod.AFCEnv(num_envs=E, resolution=res, reynolds=100, exact_sums=exact, init_time=init_time, upleg=upleg)
    assert g.upleg() == (upleg if upleg else (3 if E < 640 else 1))
    if start is not None:
        g.set_reset_state(start["u"], start["v"], start["p"])
        g.reset()
    rec = []
    for _ in range(N):
        obs, rew, done, ns = g.agent_step(rng.uniform(-1, 1, (E, 2)).astype(np.float32))
        rec.append((obs.copy(), rew.copy(), ns.copy()))
    st = [g.get_state(e) for e in range(0, E, max(1, E // 8))]
    h = g.mg_histogram()
    g.close()
    return rec, st, h


def _same_run(a, b, bits=True):
    (ra, sa, ha), (rb, sb, hb) = a, b
    assert np.array_equal(ha, hb)
    for (o1, w1, n1), (o2, w2, n2) in zip(ra, rb):
        assert np.array_equal(n1, n2)
        if bits:
            assert np.array_equal(o1.view(np.uint32), o2.view(np.uint32)) and np.array_equal(w1.view(np.uint32), w2.view(np.uint32))
        else:
            assert np.allclose(o1, o2, rtol=5e-5, atol=5e-5) and np.allclose(w1, w2, rtol=5e-5, atol=5e-5)
    for x, y in zip(sa, sb):
        for f1, f2, tol in zip(x, y, (5e-6, 5e-6, 5e-4 * max(1.0, float(np.abs(x[2]).max())))):
            if bits:
                assert np.array_equal(f1.view(np.uint32), f2.view(np.uint32))
            else:
                assert float(np.abs(f1 - f2).max()) <= tol


@pytest.mark.parametrize("res", [8, 16])
def test_uplegs_agree(envmod, res):
    """afc_config.upleg: the strip pipeline (1: afc::prod), the barrier form (2: afc::diag) and the four sweeps marching down
    the rows (3: afc::lat, what launches of few envs per CU run by default) do the same arithmetic in the same order: agent
    steps through the impulsive start (several V-cycles per solve: the lazy-residual replay runs too) with 40 different
    actions, in parity mode bit for bit -- observations, iteration histogram, fields.  In the default mode the three cut
    the level into different pieces, so the partial sums of r.r and sum(p) are added in another order: equal iteration
    counts, results within the default mode's tolerance (as for the other strip shapes, test_experiment_paths_...)."""
    for exact in (1, 0):
        ref = _run_upleg(envmod, 1, exact, res, 40)
        assert ref[2][1:].sum() > 0
        for u in (3, 2, 0):
            _same_run(ref, _run_upleg(envmod, u, exact, res, 40), bits=(exact == 1))


def test_rows_march_at_one_env_per_cu_is_deterministic_and_equals_the_barrier_form(envmod):
    """race detector for mg_up_tr (the march, the filling wave and the two exiting waves hand ring rows over through three
    counters in LDS) under the load it ships for: 256 envs = one per CU, random actions, 6 agent steps from the developed
    wake -- two runs must agree bit for bit (a timing-dependent hazard would not), and in parity mode the march down the
    rows must agree bit for bit with the barrier form; then the same at 1024 envs (upleg = 3 forced: four per CU, the
    occupancy it is NOT the default for).  tests/test_tmarch_schedule_model.py: the schedule itself on the CPU."""
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "wake_res16_re100.npz"))
    a = _run_upleg(envmod, 0, 0, 16, 256, N=6, init_time=1.0, start=z)
    _same_run(a, _run_upleg(envmod, 0, 0, 16, 256, N=6, init_time=1.0, start=z))
    _same_run(_run_upleg(envmod, 3, 1, 16, 256, N=4, init_time=1.0, start=z), _run_upleg(envmod, 2, 1, 16, 256, N=4, init_time=1.0, start=z))
    b = _run_upleg(envmod, 3, 0, 16, 1024, N=4, init_time=1.0, start=z)
    _same_run(b, _run_upleg(envmod, 3, 0, 16, 1024, N=4, init_time=1.0, start=z))
    _same_run(_run_upleg(envmod, 3, 1, 16, 1024, N=3, init_time=1.0, start=z), _run_upleg(envmod, 1, 1, 16, 1024, N=3, init_time=1.0, start=z))
