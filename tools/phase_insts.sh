#!/bin/bash
# per-phase instruction counters of one sim-step (wave-instructions per env), GPU box:  tools/phase_insts.sh [out]
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=${1:-$R/gpurun_out/phase_insts.txt}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pt_sq -- python3 $R/tools/phase_traffic.py run > /dev/null 2>&1
cd $R
python3 tools/phase_traffic.py counters gpurun_out/pt_sq > $O
rm -rf gpurun_out/pt_sq
cat $O
