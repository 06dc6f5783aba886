"""HBM bytes per phase of one sim-step, by differencing launches that leave the step early.
Run twice under rocprofv3 (--pmc FETCH_SIZE, --pmc WRITE_SIZE); then `phase_traffic.py report <dirF> <dirW> [res]`.
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out_f -- python3 tools/phase_traffic.py run [res]
res = 16 (256x128, Re 100; default), 24 (384x192, Re 500, init.bdim) or 32 (512x256, Re 400)."""
import csv
import glob
import os
import sys

import numpy as np

STOPS = [1, 2, 3, 4, 5, 6, 9, 10, 11, 12, 13, 14, 0]
NAMES = ["advdif", "forcing+setbc", "head(down0)", "coarse levels", "up0 (L0 fill/march/exit)", "project tail"]
REP = 3
E = 1024


GRIDS = {16: ("wake_res16_re100.npz", 100), 24: ("init_bdim_res24.npz", 500), 32: ("wake_res32_re400.npz", 400)}


def run(res=16):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from rlfluidcontrol_amd import build, env
    build.build()
    name, Re = GRIDS[res]
    z = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", name))
    for st in STOPS:
        g = env.AFCEnv(num_envs=E, resolution=res, reynolds=Re, dbg=st << 12)
        g.set_reset_state(z["u"], z["v"], z["p"])
        g.reset()
        xi = np.zeros((E, 2), np.float32)
        for _ in range(REP):
            g.update2(xi)
        g.close()


def series(d, counter):
    rows = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_sim_step" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    rows.sort()
    agg = {}
    for did, v in rows:
        agg[did] = agg.get(did, 0.0) + v
    vals = [agg[k] for k in sorted(agg)]
    assert len(vals) == REP * len(STOPS), len(vals)
    return [min(vals[i * REP:(i + 1) * REP][1:]) for i in range(len(STOPS))]  # skip each env's first launch


def report(df, dw, res=16):
    f = [2 * 1024 * v for v in series(df, "FETCH_SIZE")]  # gfx950: x2, KiB
    w = [1024 * v for v in series(dw, "WRITE_SIZE")]
    cells = E * (16 * res) * (8 * res)
    prev_f = prev_w = 0.0
    print("%-34s %10s %10s   (floats per cell)" % ("phase", "read", "write"))
    for i, st in enumerate(STOPS):
        if st == 0:
            print("%-34s %10.2f %10.2f" % ("whole step", f[i] / cells / 4, w[i] / cells / 4))
            continue
        half, ph = divmod(st - 1, 8)
        print("%-34s %10.2f %10.2f" % ("half %d %s" % (half, NAMES[ph]), (f[i] - prev_f) / cells / 4, (w[i] - prev_w) / cells / 4))
        prev_f, prev_w = f[i], w[i]


def report_insts(d):
    """per-phase vector-memory wave-instructions per env (one SQ pass: SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_INSTS_LDS)"""
    names = ["SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_VALU", "SQ_INSTS_LDS"]
    cols = [series(d, c) for c in names]
    prev = [0.0] * len(names)
    print("%-34s %10s %10s %10s %10s   (wave-instructions per env)" % ("phase", "vmem rd", "vmem wr", "valu", "lds"))
    for i, st in enumerate(STOPS):
        cur = [c[i] / E for c in cols]
        if st == 0:
            print("%-34s " % "whole step" + " ".join("%10.0f" % v for v in cur))
            continue
        half, ph = divmod(st - 1, 8)
        print("%-34s " % ("half %d %s" % (half, NAMES[ph])) + " ".join("%10.0f" % (a - b) for a, b in zip(cur, prev)))
        prev = cur


def report_counters(d):
    """every counter found in the pass, per phase, per env (generic form of report_insts)"""
    names = set()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_sim_step" in r["Kernel_Name"]:
                names.add(r["Counter_Name"])
    names = sorted(names)
    cols = [series(d, c) for c in names]
    prev = [0.0] * len(names)
    print("%-34s " % "phase" + " ".join("%22s" % n for n in names) + "   (per env)")
    for i, st in enumerate(STOPS):
        cur = [c[i] / E for c in cols]
        if st == 0:
            print("%-34s " % "whole step" + " ".join("%22.0f" % v for v in cur))
            continue
        half, ph = divmod(st - 1, 8)
        print("%-34s " % ("half %d %s" % (half, NAMES[ph])) + " ".join("%22.0f" % (a - b) for a, b in zip(cur, prev)))
        prev = cur


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 16)
    elif sys.argv[1] == "insts":
        report_insts(sys.argv[2])
    elif sys.argv[1] == "counters":
        report_counters(sys.argv[2])
    else:
        report(sys.argv[2], sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 16)
