"""Turns rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py` into HBM bytes per launch of
the step kernel, corrected as /opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes:
  FETCH_SIZE, WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide
  (16 B/lane) coalesced streaming reads -> doubled (an upper estimate for our mixed 4/8/16-byte
  loads; WRITE_SIZE is exact for 16 B/lane stores).  Separate passes (TCC has 4 slots).
usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [res]"""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_hash  # noqa: E402  (ties the figure to the kernel sources it was measured on)


def steady(path, counter):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
            if "k_agent_step" in r["Kernel_Name"] and r["Counter_Name"] == counter]
    vals.sort()
    return vals[len(vals) // 2] if len(vals) < 3 else sorted(vals)[0:len(vals) - 1][len(vals) // 2 - 1]


fetch_kib = steady(sys.argv[1], "FETCH_SIZE")
write_kib = steady(sys.argv[2], "WRITE_SIZE")
out = {"kernel": "k_agent_step", "fetch_size_kib_raw": fetch_kib, "write_size_kib": write_kib,
       "fetch_bytes_corrected_x2": 2 * fetch_kib * 1024, "write_bytes": write_kib * 1024,
       "hbm_bytes_per_launch": 2 * fetch_kib * 1024 + write_kib * 1024,
       "kernel_src_sha16": kernel_source_hash(),
       "grid": "%dx%d" % (16 * int(sys.argv[4]), 8 * int(sys.argv[4])) if len(sys.argv) > 4 else "256x128",
       "note": "steady launches (16 sim-steps x 1024 envs); FETCH_SIZE doubled per the gfx950 correction "
               "(upper estimate: part of our loads are 4- and 8-byte wide, for which the counter is uncalibrated)"}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(out)
