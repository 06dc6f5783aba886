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
HBM_PEAK = 8.0e12  # MI355X_MICROARCH.md: HBM3E spec peak (6.29e12 measured copy ceiling)


def _cpu_worker(args):
    res, reynolds, seconds, seed = args
    sys.path.insert(0, ROOT)
    from oracle import oracle_py
    o = oracle_py.Oracle(resolution=res, Re=reynolds)
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


def cpu_baseline(res, reynolds, seconds=12.0):
    """oracle (kind 'port') on all host cores, one independent env per core"""
    import multiprocessing as mp
    from oracle import oracle_py
    oracle_py.build()
    cores = os.cpu_count() or 1
    ctx = mp.get_context("spawn")
    with ctx.Pool(cores) as pool:
        out = pool.map(_cpu_worker, [(res, reynolds, seconds, 100 + i) for i in range(cores)])
    total = sum(n / t for n, t in out)
    one = max(n / t for n, t in out)
    return {"value": round(total, 2), "unit": "env-sim-steps/s", "cores": cores, "kind": "port",
            "per_core": round(total / cores, 2), "best_core": round(one, 2),
            "sample": "%d independent envs (one per core) of the %dx%d Re=%d case, %.0f s each, random actions "
                      "re-drawn every 16 steps, from the free stream" % (cores, 16 * res, 8 * res, reynolds, seconds)}


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
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # the CPU baseline runs in child processes: start them BEFORE this process touches the GPU
    cpu = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.res, args.reynolds)
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

    def agent_step():
        action.uniform_(-1.0, 1.0, generator=gen)
        g.agent_step_device(action.data_ptr(), obs.data_ptr(), rew.data_ptr(), done.data_ptr(), nst.data_ptr())
        if world > 1:  # the only collective on the path: (obs, reward, done) to every rank
            packed[:, 0:2] = obs
            packed[:, 2] = rew
            packed[:, 3] = done.float()
            dist.all_gather_into_tensor(gathered, packed)
        g.reset(done.data_ptr())  # restart finished episodes (mask on device)

    warm_sum = torch.zeros((), dtype=torch.int64, device=dev)
    for _ in range(args.warmup):
        agent_step()
        warm_sum += nst.sum()  # same torch ops as the timed loop (their first call loads a code object)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    sim_steps = torch.zeros((), dtype=torch.int64, device=dev)
    t0 = time.perf_counter()
    for k in range(args.steps):
        action.uniform_(-1.0, 1.0, generator=gen)
        evs[k][0].record(stream)
        g.agent_step_device(action.data_ptr(), obs.data_ptr(), rew.data_ptr(), done.data_ptr(), nst.data_ptr())
        evs[k][1].record(stream)
        sim_steps += nst.sum()
        if world > 1:
            packed[:, 0:2] = obs
            packed[:, 2] = rew
            packed[:, 3] = done.float()
            dist.all_gather_into_tensor(gathered, packed)
        g.reset(done.data_ptr())
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kern_ms = [a.elapsed_time(b) for a, b in evs]
    my_sim = int(sim_steps.item())
    diverged = int(g.diverged().sum())
    it, _ = g.solver_stats()

    tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    tot = torch.tensor([my_sim], device=dev, dtype=torch.float64)
    if dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    elapsed = float(tmax.item())
    total_sim = float(tot.item())

    if rank == 0:
        cells = (g.n - 2) * (g.m - 2)
        sim_per_launch = my_sim / args.steps
        avg_kern_s = float(np.mean(kern_ms)) * 1e-3
        achieved = ALGO_BYTES_PER_CELL_STEP * cells * sim_per_launch / avg_kern_s
        line = {
            "metric": "env-steps/sec, 256x128 Re=100 cylinder, 1024 envs @ 1/2/4/8 GPUs"
                      if (args.res, args.reynolds, E) == (16, 100, 1024) else
                      "env-steps/sec, %dx%d Re=%d cylinder, %d envs/GPU" % (16 * args.res, 8 * args.res, args.reynolds, E),
            "value": round(total_sim / elapsed, 1),
            "unit": "env-sim-steps/s (1 env-step = one AFCCylinder.update2; 16 per agent step)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic (init: %s; actions U(-1,1)^2 per env per agent step, Philox seed 1234+rank)" % init_used,
            "config": {"workload": "BASELINE.json configs[1]" if (args.res, args.reynolds, E) == (16, 100, 1024) else "custom",
                       "envs_per_gpu": E, "grid": "%dx%d" % (16 * args.res, 8 * args.res), "reynolds": args.reynolds,
                       "sim_steps_per_agent_step": 16, "parallelism": "env-sharded x%d" % world,
                       "gs_order": "lexicographic (reference order)", "mg_iters_last": [int(it[:, 0].max()), int(it[:, 1].max())]},
            "agent_steps_per_s": round(total_sim / 16.0 / elapsed, 2),
            "diverged_envs": diverged,
            "roofline": {"bound": "hbm", "achieved": round(achieved / 1e9, 2), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK, 5), "traffic": None,
                         "kernel": "k_agent_step", "avg_launch_ms": round(1e3 * avg_kern_s, 3),
                         "algorithmic_bytes_per_launch": ALGO_BYTES_PER_CELL_STEP * cells * sim_per_launch,
                         "frac_of_measured_copy_ceiling_6.29TBs": round(achieved / 6.29e12, 5)},
        }
        # HBM bytes per launch from the PMC passes committed under profiles/ (rocprofv3 cannot be
        # driven from inside this process); only valid for the exact workload they were taken on
        tpath = os.path.join(ROOT, "profiles", "r1_hbm_traffic.json")
        if os.path.exists(tpath) and (args.res, args.reynolds, E) == (16, 100, 1024):
            try:
                tj = json.load(open(tpath))
                line["roofline"]["traffic"] = tj["hbm_bytes_per_launch"]
                line["roofline"]["traffic_source"] = "profiles/r1_hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)"
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
