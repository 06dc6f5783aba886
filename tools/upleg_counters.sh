#!/bin/bash
# tools/upleg_counters.sh: SQ counters of the isolated up-leg kernels of tools/upleg_bench.hip at 1024 workgroups (four per CU, all
# in the same phase): mg_up_smooth (k<0>), mg_up_pipe (k<1>), mg_up_tr (k<2>) -- per workgroup and launch of 8 calls.  GPU box, from the
# repo root, with tools/bin/upleg_bench built (see the head of tools/upleg_bench.hip).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/upleg_ctr
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $O/p1 -- $R/tools/bin/upleg_bench 128 0 1 1 1 1024 > $O/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/p2 -- $R/tools/bin/upleg_bench 128 0 1 1 1 1024 > $O/p2.log 2>&1
cd $R
python3 - <<'PY'
import csv,glob,collections
for p in ("p1","p2"):
    files=glob.glob("gpurun_out/upleg_ctr/%s/**/*counter_collection.csv"%p, recursive=True)
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        for row in csv.DictReader(open(f)):
            agg[row["Kernel_Name"][:30]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k,v in agg.items():
        print(p,k, {c: round(sum(x)/len(x)/1024) for c,x in v.items()}, "(per workgroup, mean over launches)")
PY
rm -rf $O/p1 $O/p2
