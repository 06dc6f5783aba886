#!/bin/bash
# per-phase SQ / instruction-cache / LDS counters of one sim-step (per env), GPU box:  tools/phase_counters.sh <outdir>
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$(realpath -m ${1:-$R/gpurun_out/phase_counters})
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $O/p1 -- python3 $R/tools/phase_traffic.py run > $O/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA --output-format csv -d $O/p2 -- python3 $R/tools/phase_traffic.py run > $O/p2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_SMEM --output-format csv -d $O/p3 -- python3 $R/tools/phase_traffic.py run > $O/p3.log 2>&1
cd $R
for p in p1 p2 p3; do python3 tools/phase_traffic.py counters $O/$p > $O/$p.txt 2>&1; rm -rf $O/$p; done
cat $O/p1.txt $O/p2.txt $O/p3.txt
