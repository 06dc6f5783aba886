#!/usr/bin/env python
"""bench.py -- env-steps/sec of the batched BDIM cylinder-wake environment on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--envs E] [--res R] [--reynolds RE] [--gs-mode 0|1]

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
GOLDEN = os.path.join(ROOT, "tests", "golden")
# committed start states: (res, Re) -> file
DEFAULT_INIT = {(16, 100): os.path.join(GOLDEN, "wake_res16_re100.npz"),
                (32, 400): os.path.join(GOLDEN, "wake_res32_re400.npz"),
                (24, 500): os.path.join(GOLDEN, "init_bdim_res24.npz")}
# BASELINE.json workloads by (res, Re, envs per GPU)
# HBM traffic per launch measured with rocprofv3 --pmc on exactly these kernel sources (tools/pmc_traffic.py), by workload
TRAFFIC_FILES = {(16, 100, 1024): "r6_hbm_traffic.json", (32, 400, 1024): "r6_hbm_traffic_cfg4.json",
                 (24, 500, 1024): "r6_hbm_traffic_native.json"}
WORKLOADS = {(16, 100, 1024): "BASELINE.json configs[1]",
             (32, 400, 1024): "BASELINE.json configs[3]",
             (24, 500, 1024): "reference-native 384x192 Re500 (init.bdim)"}


def kernel_source_hash():
    """sha256 (first 16 hex digits) over the CODE of the kernel sources -- comments and white space stripped, so that
    editing a comment does not cut a committed PMC traffic figure loose from the code it was measured on"""
    import hashlib
    import re
    h = hashlib.sha256()
    d = os.path.join(ROOT, "rlfluidcontrol_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".hip", ".cpp")):
            txt = open(os.path.join(d, f), "r", errors="replace").read()
            txt = re.sub(r"/\*.*?\*/", " ", txt, flags=re.S)      # block comments
            txt = re.sub(r"//[^\n]*", " ", txt)                   # line comments (none of the sources has `//` in a string)
            txt = re.sub(r"\s+", " ", txt)
            h.update(f.encode())
            h.update(txt.encode())
    return h.hexdigest()[:16]


class _StubEnv:
    """--dry-run-cpu: stands in for AFCEnv on a box without a GPU (torch CPU tensors, a fixed sleep per agent step)"""
    n, m = 258, 130

    def __init__(self, num_envs, rank):
        import torch
        self.E, self.rank, self.torch = num_envs, rank, torch
        self.hist = np.zeros(8, np.int64)

    def step(self, action, obs, rew, done, nst):
        time.sleep(0.002 * (1 + self.rank))  # ranks differ: the slowest one defines the step
        obs.copy_(action * 0.5 + self.rank)
        rew.copy_(-obs[:, 1])
        done.zero_()
        nst.fill_(16)
        self.hist[0] += 2 * 16 * self.E

    def diverged(self):
        return np.zeros(self.E, np.uint8)

    def solver_stats(self):
        return np.ones((self.E, 2), np.int32), np.zeros((self.E, 2), np.float32)

    def mg_histogram(self, clear=False):
        h = self.hist.copy()
        if clear:
            self.hist[:] = 0
        return h

    def close(self):
        pass


class _CpuEvent:
    def __init__(self):
        self.t = 0.0

    def record(self, stream=None):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return 1e3 * (other.t - self.t)

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
    ap.add_argument("--steps", type=int, default=64, help="timed agent steps per pass (SURVEY 8(d): >= 64)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--envs", type=int, default=1024, help="envs per GPU")
    ap.add_argument("--res", type=int, default=16, help="resolution: grid = 16*res x 8*res")
    ap.add_argument("--reynolds", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stagger-us", type=int, default=0)
    ap.add_argument("--dbg", type=int, default=0, help="experiment switches (development only)")
    ap.add_argument("--init", default=None,
                    help="state (.npz with u, v, p) to start from; default: the committed state of the workload "
                         "(developed wake for 256x128 Re 100 and 512x256 Re 400, the reference's init.bdim for 384x192 "
                         "Re 500), else the free stream")
    ap.add_argument("--upleg", type=int, default=0,
                    help="afc_config.upleg: 0 by occupancy (default), 1 strip pipeline, 2 barrier form, 3 marching down the rows")
    ap.add_argument("--gs-mode", type=int, default=0,
                    help="0 = the reference's lexicographic Gauss-Seidel (headline); 1 = red-black perf mode (separate line)")
    ap.add_argument("--dry-run-cpu", action="store_true",
                    help="exercise the launch / rendezvous / gather / reporting path on CPU (gloo) with a stub in place "
                         "of the HIP env: for tests of the N>1 path on boxes without GPUs; the numbers mean nothing")
    ap.add_argument("--repeats", type=int, default=3, help="timed passes of --steps agent steps; the median is reported")
    args = ap.parse_args()
    if args.init is None:
        args.init = DEFAULT_INIT.get((args.res, args.reynolds))

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
    dry = args.dry_run_cpu
    if not dry and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback in the product path)")
    dist = None
    if dry:
        dev = torch.device("cpu")
        if world > 1:
            import torch.distributed as dist
            dist.init_process_group("gloo")
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        if world > 1:
            import torch.distributed as dist
            dist.init_process_group("nccl", device_id=dev)

    def sync():
        if not dry:
            torch.cuda.synchronize()

    def new_event():
        return _CpuEvent() if dry else torch.cuda.Event(enable_timing=True)

    E = args.envs
    init_used = "free stream"
    if dry:
        g = _StubEnv(E, rank)
        stream = None
        eff_dbg = 0
    else:
        from rlfluidcontrol_amd import build, env as envmod
        build.build()
        g = envmod.AFCEnv(num_envs=E, resolution=args.res, reynolds=args.reynolds, device=local_rank,
                          stagger_us=args.stagger_us, dbg=args.dbg, gs_mode=args.gs_mode, upleg=args.upleg)
        eff_dbg = int(g.cfg.reserved[0])  # includes AFC_DBG from the environment: recorded in the line
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
        if dry:
            g.step(action, obs, rew, done, nst)
        else:
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
        if not dry:
            g.reset(done.data_ptr())  # restart finished episodes (mask on device)

    warm_sum = torch.zeros((), dtype=torch.int64, device=dev)
    for _ in range(args.warmup):
        agent_step()
        warm_sum += nst.sum()  # same torch ops as the timed loop (their first call loads a code object)
    sync()
    g.mg_histogram(clear=True)

    # device-to-device copy ceiling of THIS box, measured in the same run: the library's hand-written 16-byte-per-lane
    # copy kernel (1 GiB, 2048 workgroups x 256 threads; afc_measure_copy_rate) is the hard ceiling fractions are
    # quoted against; torch's Tensor.copy_ of the same size (a softer figure) is reported beside it
    copy_bw = torch_copy_bw = None
    if rank == 0 and not dry:
        import ctypes
        nb = 1 << 30
        gbs = ctypes.c_float(0.0)
        try:
            if envmod.load_library().afc_measure_copy_rate(local_rank, nb, 5, ctypes.byref(gbs)) == 0:
                copy_bw = float(gbs.value) * 1e9
        except AttributeError:  # an older build under tools/ab.sh
            pass
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
        torch_copy_bw = 2.0 * nb / best
        del a, b
        torch.cuda.empty_cache()

    passes = []  # (elapsed seconds of K steps on this rank, kernel ms list, gather ms list, sim-steps)
    for _rep in range(max(1, args.repeats)):
        if dist:
            dist.barrier()
        sync()
        evs = [(new_event(), new_event()) for _ in range(args.steps)]
        gevs = [(new_event(), new_event()) for _ in range(args.steps)]
        sim_steps = torch.zeros((), dtype=torch.int64, device=dev)
        t0 = time.perf_counter()
        for k in range(args.steps):
            agent_step(evs[k], gevs[k] if world > 1 else None)
            sim_steps += nst.sum()
        sync()
        if dist:
            dist.barrier()
        sync()
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
        std = (args.res, args.reynolds, E) == (16, 100, 1024) and args.gs_mode == 0
        workload = WORKLOADS.get((args.res, args.reynolds, E), "custom")
        if args.gs_mode:
            workload += " -- PERF MODE smoother (red-black Gauss-Seidel: not the reference's iterates, see DESIGN)"
        if dry:
            workload = "dry run on CPU (stub env): launch / gather / reporting path only"
        line = {
            "metric": "env-steps/sec, 256x128 Re=100 cylinder, 1024 envs @ 1/2/4/8 GPUs" if std else
                      "env-steps/sec, %dx%d Re=%d cylinder, %d envs/GPU" % (16 * args.res, 8 * args.res, args.reynolds, E),
            "value": round(total_sim / elapsed, 1),
            "unit": "env-sim-steps/s (1 env-step = one AFCCylinder.update2; 16 per agent step)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic (init: %s; actions U(-1,1)^2 per env per agent step, Philox seed 1234+rank)" % init_used,
            "config": {"workload": workload,
                       "envs_per_gpu": E, "grid": "%dx%d" % (16 * args.res, 8 * args.res), "reynolds": args.reynolds,
                       "sim_steps_per_agent_step": 16, "parallelism": "env-sharded x%d" % world,
                       "gs_order": "red-black (perf mode)" if args.gs_mode else "lexicographic (reference order)",
                       "upleg": (g.upleg() if hasattr(g, "upleg") else None),
                       "dbg": eff_dbg, "kernel_src_sha16": kernel_source_hash(), "mg_iters_last": [int(it[:, 0].max()), int(it[:, 1].max())],
                       "mg_iters_histogram": {("%d" % (k + 1)) if k < 7 else ">=8": int(hist[k]) for k in range(8) if hist[k]}},
            "timing": {"repeats": len(passes), "reported": "median pass", "ms_per_step_all_passes":
                       [round(1e3 * q[0] / args.steps, 3) for q in passes], "ms_per_step_per_rank": per_rank_ms,
                       "gather_ms_per_step": round(float(np.mean(gather_ms)), 4) if gather_ms else 0.0,
                       "gather_share_of_step": round(float(np.mean(gather_ms)) / (1e3 * elapsed / args.steps), 5) if gather_ms else 0.0},
            "placement": (g.placement_info() if hasattr(g, "placement_info") else None),
            "agent_steps_per_s": round(total_sim / 16.0 / elapsed, 2),
            "diverged_envs": diverged,
            "roofline": {"bound": "hbm", "achieved": round(achieved / 1e9, 2), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK, 5), "traffic": None,
                         "kernel": "k_agent_step", "avg_launch_ms": round(1e3 * avg_kern_s, 3),
                         "algorithmic_bytes_per_launch": ALGO_BYTES_PER_CELL_STEP * cells * sim_per_launch,
                         "copy_ceiling_measured_GBs": round(copy_bw / 1e9, 1) if copy_bw else None,
                         "frac_of_measured_copy_ceiling": round(achieved / copy_bw, 5) if copy_bw else None,
                         "copy_ceiling_kernel": "afc_measure_copy_rate: hand-written float4 copy, 1 GiB, best of plain / non-temporal x 2048 / 8192 workgroups",
                         "torch_copy_GBs": round(torch_copy_bw / 1e9, 1) if torch_copy_bw else None},
        }
        # HBM bytes per launch from the PMC passes committed under profiles/ (rocprofv3 cannot be driven from inside
        # this process): attached only for the exact workload AND the exact kernel sources they were taken on
        tname = TRAFFIC_FILES.get((args.res, args.reynolds, E))
        tpath = os.path.join(ROOT, "profiles", tname) if tname else None
        if tpath and os.path.exists(tpath) and args.gs_mode == 0 and not dry:
            try:
                tj = json.load(open(tpath))
                if tj.get("kernel_src_sha16") == kernel_source_hash():
                    line["roofline"]["traffic"] = tj["hbm_bytes_per_launch"]
                    line["roofline"]["traffic_source"] = "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, same kernel sources)" % tname
                else:
                    line["roofline"]["traffic_source"] = "none: profiles/%s was measured on other kernel sources (%s)" % (tname, tj.get("kernel_src_sha16"))
            except Exception:
                pass
        if (args.res, args.reynolds) == (32, 400):
            line["roofline"]["note"] = ("176 B per cell and sim-step is the contract figure for every grid; at 512x256 "
                                        "level 1 no longer fits on chip and streams (SURVEY 8(d): ~185 B per cell)")
        if cpu is not None:
            line["cpu_baseline"] = cpu
        print(json.dumps(line))
    g.close()
    if dist:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    main()
