"""Median per-launch value of every counter rocprofv3 collected for the step kernel.
usage: pmc_summary.py <dir with *counter_collection.csv> [kernel substring]"""
import csv
import glob
import json
import sys
from collections import defaultdict

pat = sys.argv[2] if len(sys.argv) > 2 else "k_agent_step"
vals = defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    per = defaultdict(float)
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (d, c), v in per.items():
        vals[c].append(v)
out = {}
for c, v in sorted(vals.items()):
    v.sort()
    out[c] = v[len(v) // 2]
print(json.dumps(out, indent=1))
