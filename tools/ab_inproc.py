"""Same-process A/B of builds of the library: every library gets its own env (1024 envs, the bench workload, the same
start state and the same actions) and the agent-step launches of the libraries ALTERNATE in blocks of a few launches,
so that clock / thermal drift of the box -- which moves whole bench runs by +-4 % (tools/ab.sh) -- hits all of them alike.
Reports the mean launch time per library over all blocks and the ratio to the first one.
    python tools/ab_inproc.py [--res 16 --reynolds 100 --envs 1024 --blocks 12 --per-block 6] libA.so libB.so ..."""
import argparse
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def load_env_module(tag, libpath):
    """a private copy of rlfluidcontrol_amd/env.py bound to `libpath` (the module caches ONE library)"""
    spec = importlib.util.spec_from_file_location("afc_env_" + tag, os.path.join(ROOT, "rlfluidcontrol_amd", "env.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    os.environ["AFC_ENV_LIB"] = os.path.abspath(libpath)
    m.load_library()
    return m


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", type=int, default=16)
    ap.add_argument("--reynolds", type=int, default=100)
    ap.add_argument("--envs", type=int, default=1024)
    ap.add_argument("--blocks", type=int, default=12)
    ap.add_argument("--per-block", type=int, default=6)
    ap.add_argument("--action-scale", type=float, default=1.0, help="actions are U(-s, s)^2 (0: no actuation, all envs alike)")
    ap.add_argument("--share-slab", action="store_true",
                    help="every instance adopts the state slab of the first one (AFC_SLAB_PTR): builds are compared on the "
                         "SAME memory -- they must be bit-identical builds, the simulation simply continues from launch to launch")
    ap.add_argument("libs", nargs="+")
    a = ap.parse_args()
    import torch
    import bench
    dev = torch.device("cuda", 0)
    init = bench.DEFAULT_INIT.get((a.res, a.reynolds))
    z = np.load(init) if init and os.path.exists(init) else None
    stream = torch.cuda.current_stream()
    envs = []
    for k, lib in enumerate(a.libs):
        # "lib.so:VAR=value,VAR2=value" sets environment variables while this library's env is created (allocation
        # experiments); lower-case names are afc_config fields instead ("lib.so:upleg=3")
        lib, _, evs = lib.partition(":")
        a.libs[k] = lib + (":" + evs if evs else "")
        cfg_over = {}
        for kv in filter(None, evs.split(",")):
            if kv.split("=")[0].islower():
                cfg_over[kv.split("=")[0]] = int(kv.split("=")[1])
            else:
                os.environ[kv.split("=")[0]] = kv.split("=")[1]
        if a.share_slab and k == 0:
            os.environ["AFC_EXPORT_SLAB"] = "1"
        if a.share_slab and k > 0:
            import ctypes
            libc = ctypes.CDLL(None)
            libc.getenv.restype = ctypes.c_char_p
            os.environ["AFC_SLAB_PTR"] = libc.getenv(b"AFC_LAST_SLAB").decode()
        m = load_env_module(str(k), lib)
        g = m.AFCEnv(num_envs=a.envs, resolution=a.res, reynolds=a.reynolds, device=0, **cfg_over)
        os.environ.pop("AFC_SLAB_PTR", None)
        if z is not None and z["u"].shape == (g.n, g.m):
            g.set_reset_state(z["u"], z["v"], z["p"])
        g.use_device_io(stream.cuda_stream)
        envs.append(g)
        for kv in filter(None, evs.split(",")):
            if not kv.split("=")[0].islower():
                os.environ.pop(kv.split("=")[0], None)
    E = a.envs
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234)
    action = torch.zeros(E, 2, device=dev)
    obs = torch.zeros(E, 2, device=dev)
    rew = torch.zeros(E, device=dev)
    done = torch.zeros(E, dtype=torch.uint8, device=dev)
    nst = torch.zeros(E, dtype=torch.int32, device=dev)

    def step(g):
        g.agent_step_device(action.data_ptr(), obs.data_ptr(), rew.data_ptr(), done.data_ptr(), nst.data_ptr())

    for g in envs:  # warm-up: the first launch runs the 149 pre-action steps
        for _ in range(3):
            action.uniform_(-1.0, 1.0, generator=gen).mul_(a.action_scale)
            step(g)
    torch.cuda.synchronize()
    ms = [[] for _ in envs]
    for b in range(a.blocks):
        order = list(range(len(envs)))
        if b % 2:
            order.reverse()
        for k in order:
            evs = []
            for _ in range(a.per_block):
                action.uniform_(-1.0, 1.0, generator=gen).mul_(a.action_scale)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                step(envs[k])
                e1.record(stream)
                evs.append((e0, e1))
            torch.cuda.synchronize()
            ms[k].append(float(np.mean([x.elapsed_time(y) for x, y in evs])))
    base = float(np.mean(ms[0]))
    for k, lib in enumerate(a.libs):
        v = np.array(ms[k])
        print("%-40s %8.3f ms per launch (min %.3f, max %.3f over %d blocks)  %+.2f %% time vs first"
              % (os.path.basename(lib), v.mean(), v.min(), v.max(), len(v), 100.0 * (v.mean() / base - 1.0)))
    for k, g in enumerate(envs if not a.share_slab else envs[:1]):
        h = g.mg_histogram()
        print("  env %d: multigrid iterations per solve, all launches: %s" % (k, {i + 1: int(v) for i, v in enumerate(h) if v}))
    for g in reversed(envs):  # (a shared slab belongs to the first instance: closed last)
        g.close()


if __name__ == "__main__":
    main()
