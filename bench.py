#!/usr/bin/env python
"""bench.py -- env-steps/sec of the batched BDIM cylinder-wake environment on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--envs E] [--res R] [--reynolds RE]

A "step" is one AGENT step of the whole batch: 16 x AFCCylinder.update2 per env (clientCFD.pde:12,
35-56) in ONE kernel launch, then (N>1) the RCCL all-gather of (obs, reward, done).  Workload at
N=1 is BASELINE.json configs[1]: 1024 envs, 256x128 grid, Re=100, fp32.  Weak scaling: every rank
owns E envs; there is no data-path collective besides the per-agent-step gather.

Prints ONE JSON line (rank 0).  `value` = env sim-steps/s over all ranks, inputs resident in HBM.
`roofline` prices the step kernel against HBM with SURVEY.md §8(d)'s ALGORITHMIC bytes:
176 B per interior cell per sim-step.  `cpu_baseline` times the CPU oracle (a literal port of the
reference's Java arithmetic, oracle/) on this box's host cores for a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_CELL_STEP = 176.0  # 44 fp32 field passes per sim-step
HBM_PEAK = 8.0e12  # MI355X_MICROARCH.md: HBM3E spec peak; the achievable copy rate of the box is measured in the run


def _cpu_worker(args):
    res, reynolds, seconds, seed, init = args
    sys.path.insert(0, ROOT)
    from oracle import oracle_py
    o = oracle_py.Oracle(resolution=res, Re=reynolds)
    if init is not None and os.path.exists(init):
        z = np.load(init)
        if z["u"].shape == (o.n, o.m):
            o.set_state(z["u"], z["v"], z["p"])  # the same developed wake the GPU run starts from
    rng = np.random.default_rng(seed)
    for _ in range(4):
        o.update2(0.0, 0.0)
    n = 0
    t0 = time.perf_counter()
    a = (0.0, 0.0)
    while time.perf_counter() - t0 < seconds:
        if n % 16 == 0:
            a = tuple(5.0 * rng.uniform(-1, 1, 2))
        o.update2(float(a[0]), float(a[1]))
        n += 1
    return n, time.perf_counter() - t0


def _usable_cores():
    """cores this process may run on (cgroup/affinity aware) and how many of them are distinct physical cores"""
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except AttributeError:
        cpus = list(range(os.cpu_count() or 1))
    phys = set()
    for c in cpus:
        try:
            with open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c) as f:
                phys.add(f.read().strip())
        except OSError:
            phys.add(str(c))
    return cpus, max(1, len(phys))


def cpu_baseline(res, reynolds, init, seconds=8.0):
    """oracle (kind 'port'): first ONE env on one core with the rest of the machine idle (the analogue of the
    reference's single-threaded client), then one independent env per physical core this process may use"""
    import multiprocessing as mp
    from oracle import oracle_py
    oracle_py.build()
    cpus, nphys = _usable_cores()
    ctx = mp.get_context("spawn")
    with ctx.Pool(1) as pool:
        n1, t1 = pool.map(_cpu_worker, [(res, reynolds, seconds, 99, init)])[0]
    with ctx.Pool(nphys) as pool:
        out = pool.map(_cpu_worker, [(res, reynolds, seconds, 100 + i, init) for i in range(nphys)])
    total = sum(n / t for n, t in out)
    return {"value": round(total, 2), "unit": "env-sim-steps/s", "cores": nphys, "kind": "port",
            "one_env_one_core_uncontended": round(n1 / t1, 2), "per_core_when_all_busy": round(total / nphys, 2),
            "logical_cpus_available": len(cpus),
            "sample": "%dx%d Re=%d case from the same developed wake as the GPU run, random actions re-drawn every 16 "
                      "steps: 1 env on 1 core for %.0f s (uncontended), then %d independent envs, one process per "
                      "physical core available to this process, %.0f s each" % (16 * res, 8 * res, reynolds, seconds,
                                                                               nphys, seconds)}


def _spawn_ranks(n, argv):
    """`python bench.py --gpus N` outside a launcher: start N ranks (one per GPU) before this process touches any
    GPU, relay their output, exit with their status."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--envs", type=int, default=1024, help="envs per GPU")
    ap.add_argument("--res", type=int, default=16, help="resolution: grid = 16*res x 8*res")
    ap.add_argument("--reynolds", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stagger-us", type=int, default=0)
    ap.add_argument("--dbg", type=int, default=0, help="experiment switches (development only)")
    ap.add_argument("--init", default=os.path.join(ROOT, "tests", "golden", "wake_res16_re100.npz"),
                    help="developed-wake state to start from (falls back to the free stream)")
    ap.add_argument("--repeats", type=int, default=3, help="timed passes of --steps agent steps; the median is reported")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us: become one (children are started before any GPU call in this process)
        raise SystemExit(_spawn_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0:
        print("bench.py: --gpus %d but the launcher started %d rank(s); reporting n_gpus=%d" % (args.gpus, world, world),
              file=sys.stderr)
    # the CPU baseline runs in child processes: start them BEFORE this process touches the GPU
    cpu = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.res, args.reynolds, args.init)
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback in the product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    from rlfluidcontrol_amd import build, env as envmod
    build.build()
    E = args.envs
    g = envmod.AFCEnv(num_envs=E, resolution=args.res, reynolds=args.reynolds, device=local_rank,
                      stagger_us=args.stagger_us, dbg=args.dbg)
    init_used = "free stream"
    if args.init and os.path.exists(args.init):
        z = np.load(args.init)
        if z["u"].shape == (g.n, g.m):
            g.set_reset_state(z["u"], z["v"], z["p"])
            init_used = os.path.basename(args.init)
    stream = torch.cuda.current_stream()
    g.use_device_io(stream.cuda_stream)

    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    action = torch.zeros(E, 2, device=dev)
    obs = torch.zeros(E, 2, device=dev)
    rew = torch.zeros(E, device=dev)
    done = torch.zeros(E, dtype=torch.uint8, device=dev)
    nst = torch.zeros(E, dtype=torch.int32, device=dev)
    packed = torch.zeros(E, 4, device=dev)
    gathered = torch.zeros(world * E, 4, device=dev) if world > 1 else None

    def agent_step(ev=None, gev=None):
        action.uniform_(-1.0, 1.0, generator=gen)
        if ev is not None:
            ev[0].record(stream)
        g.agent_step_device(action.data_ptr(), obs.data_ptr(), rew.data_ptr(), done.data_ptr(), nst.data_ptr())
        if ev is not None:
            ev[1].record(stream)
        if world > 1:  # the only collective on the path: (obs, reward, done) to every rank
            if gev is not None:
                gev[0].record(stream)
            packed[:, 0:2] = obs
            packed[:, 2] = rew
            packed[:, 3] = done.float()
            dist.all_gather_into_tensor(gathered, packed)
            if gev is not None:
                gev[1].record(stream)
        g.reset(done.data_ptr())  # restart finished episodes (mask on device)

    warm_sum = torch.zeros((), dtype=torch.int64, device=dev)
    for _ in range(args.warmup):
        agent_step()
        warm_sum += nst.sum()  # same torch ops as the timed loop (their first call loads a code object)
    torch.cuda.synchronize()
    g.mg_histogram(clear=True)

    # device-to-device copy ceiling of THIS box, measured in the same run (1 GiB float4 copy, read + write bytes)
    copy_bw = None
    if rank == 0:
        nb = 1 << 30
        a = torch.empty(nb // 4, dtype=torch.float32, device=dev).normal_()
        b = torch.empty_like(a)
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            b.copy_(a)
            e1.record(stream)
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e-3)
        copy_bw = 2.0 * nb / best
        del a, b
        torch.cuda.empty_cache()

    passes = []  # (elapsed seconds of K steps on this rank, kernel ms list, gather ms list, sim-steps)
    for _rep in range(max(1, args.repeats)):
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        gevs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        sim_steps = torch.zeros((), dtype=torch.int64, device=dev)
        t0 = time.perf_counter()
        for k in range(args.steps):
            agent_step(evs[k], gevs[k] if world > 1 else None)
            sim_steps += nst.sum()
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        tm = torch.tensor([el], device=dev, dtype=torch.float64)
        ts = torch.tensor([float(sim_steps.item())], device=dev, dtype=torch.float64)
        if dist:
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)  # the slowest rank defines the step
            dist.all_reduce(ts, op=dist.ReduceOp.SUM)
        passes.append((float(tm.item()), [a.elapsed_time(b) for a, b in evs],
                       [a.elapsed_time(b) for a, b in gevs] if world > 1 else [], float(ts.item()), el,
                       int(sim_steps.item())))
    order = sorted(range(len(passes)), key=lambda i: passes[i][0])
    med = passes[order[len(order) // 2]]
    elapsed, kern_ms, gather_ms, total_sim, my_elapsed, my_sim = med
    diverged = int(g.diverged().sum())
    it, _ = g.solver_stats()
    hist = g.mg_histogram()
    per_rank_ms = [round(1e3 * my_elapsed / args.steps, 3)]
    if dist:
        lst = [None] * world
        dist.all_gather_object(lst, per_rank_ms[0])
        per_rank_ms = lst

    if rank == 0:
        cells = (g.n - 2) * (g.m - 2)
        sim_per_launch = my_sim / args.steps
        avg_kern_s = float(np.mean(kern_ms)) * 1e-3
        achieved = ALGO_BYTES_PER_CELL_STEP * cells * sim_per_launch / avg_kern_s
        std = (args.res, args.reynolds, E) == (16, 100, 1024)
        line = {
            "metric": "env-steps/sec, 256x128 Re=100 cylinder, 1024 envs @ 1/2/4/8 GPUs" if std else
                      "env-steps/sec, %dx%d Re=%d cylinder, %d envs/GPU" % (16 * args.res, 8 * args.res, args.reynolds, E),
            "value": round(total_sim / elapsed, 1),
            "unit": "env-sim-steps/s (1 env-step = one AFCCylinder.update2; 16 per agent step)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic (init: %s; actions U(-1,1)^2 per env per agent step, Philox seed 1234+rank)" % init_used,
            "config": {"workload": "BASELINE.json configs[1]" if std else "custom",
                       "envs_per_gpu": E, "grid": "%dx%d" % (16 * args.res, 8 * args.res), "reynolds": args.reynolds,
                       "sim_steps_per_agent_step": 16, "parallelism": "env-sharded x%d" % world,
                       "gs_order": "lexicographic (reference order)", "mg_iters_last": [int(it[:, 0].max()), int(it[:, 1].max())],
                       "mg_iters_histogram": {("%d" % (k + 1)) if k < 7 else ">=8": int(hist[k]) for k in range(8) if hist[k]}},
            "timing": {"repeats": len(passes), "reported": "median pass", "ms_per_step_all_passes":
                       [round(1e3 * q[0] / args.steps, 3) for q in passes], "ms_per_step_per_rank": per_rank_ms,
                       "gather_ms_per_step": round(float(np.mean(gather_ms)), 4) if gather_ms else 0.0,
                       "gather_share_of_step": round(float(np.mean(gather_ms)) / (1e3 * elapsed / args.steps), 5) if gather_ms else 0.0},
            "agent_steps_per_s": round(total_sim / 16.0 / elapsed, 2),
            "diverged_envs": diverged,
            "roofline": {"bound": "hbm", "achieved": round(achieved / 1e9, 2), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK, 5), "traffic": None,
                         "kernel": "k_agent_step", "avg_launch_ms": round(1e3 * avg_kern_s, 3),
                         "algorithmic_bytes_per_launch": ALGO_BYTES_PER_CELL_STEP * cells * sim_per_launch,
                         "copy_ceiling_measured_GBs": round(copy_bw / 1e9, 1) if copy_bw else None,
                         "frac_of_measured_copy_ceiling": round(achieved / copy_bw, 5) if copy_bw else None},
        }
        # HBM bytes per launch from the PMC passes committed under profiles/ (rocprofv3 cannot be
        # driven from inside this process); only valid for the exact workload they were taken on
        for tag in ("r2", "r1"):
            tpath = os.path.join(ROOT, "profiles", "%s_hbm_traffic.json" % tag)
            if os.path.exists(tpath) and std:
                try:
                    tj = json.load(open(tpath))
                    line["roofline"]["traffic"] = tj["hbm_bytes_per_launch"]
                    line["roofline"]["traffic_source"] = "profiles/%s_hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)" % tag
                    break
                except Exception:
                    pass
        if cpu is not None:
            line["cpu_baseline"] = cpu
        print(json.dumps(line))
    g.close()
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
