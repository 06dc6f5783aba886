"""average duration of the steady k_agent_step launches (16 sim-steps) in a rocprofv3 kernel trace"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_agent_step" in r["Kernel_Name"]]
d = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6 for r in rows)
steady = [x for x in d if x < 2.5 * d[0]]
print("k_agent_step launches %d, steady %d: avg %.3f ms min %.3f max %.3f (warm-up launches: %s ms)" % (
    len(d), len(steady), sum(steady) / len(steady), steady[0], steady[-1], ["%.0f" % x for x in d if x >= 2.5 * d[0]]))
