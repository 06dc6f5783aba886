#!/bin/bash
# Regenerates every artefact under profiles/ on a GPU box (run through gpurun from the repo root):
#   tools/refresh_profiles.sh <tag>      e.g. r1   -> gpurun_out/profiles_<tag>/
# bench line, rocprofv3 kernel stats of the same command, steady launch average, HBM traffic (two PMC
# passes), SQ counters, per-phase time (1024 and 1 envs) and per-phase HBM traffic.
TAG=${1:-r2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/profiles_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- $B --no-cpu-baseline --steps 6 --warmup 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- $B --no-cpu-baseline --steps 6 --warmup 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $O/pmc_sq -- $B --no-cpu-baseline --steps 6 --warmup 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pt_f -- python3 $R/tools/phase_traffic.py run > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pt_w -- python3 $R/tools/phase_traffic.py run > /dev/null 2>&1
cd $R
python3 tools/pmc_traffic.py $(find $O/pmc_f -name "*counter_collection.csv" | head -1) $(find $O/pmc_w -name "*counter_collection.csv" | head -1) $O/${TAG}_hbm_traffic.json > /dev/null
cp $O/${TAG}_hbm_traffic.json $R/profiles/${TAG}_hbm_traffic.json   # bench.py quotes it as roofline.traffic
timeout 600 python bench.py > $O/${TAG}_bench.json 2> $O/bench.err
(cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B --no-cpu-baseline > $O/${TAG}_bench_under_rocprof.json 2>/dev/null)
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/${TAG}_bench_kernel_stats.csv
python3 tools/steady_kernel_avg.py $(find $O/stats -name "*kernel_trace.csv" | head -1) > $O/${TAG}_steady_launch.txt
python3 tools/pmc_summary.py $O/pmc_sq > $O/${TAG}_sq_counters.json
python3 tools/phase_traffic.py report $O/pt_f $O/pt_w > $O/${TAG}_phase_traffic.txt
timeout 200 python tools/phase_profile.py 1024 16 8 > $O/${TAG}_phase_profile_E1024.txt
timeout 200 python tools/phase_profile.py 1 16 8 > $O/${TAG}_phase_profile_E1.txt
timeout 300 python tools/phase_times.py 1024 16 > $O/${TAG}_phase_times_E1024.txt 2>&1
timeout 300 python tools/phase_times.py 1 16 > $O/${TAG}_phase_times_E1.txt 2>&1
rm -rf $O/stats $O/pmc_f $O/pmc_w $O/pmc_sq $O/pt_f $O/pt_w
bash tools/phase_insts.sh $O/${TAG}_phase_insts.txt > /dev/null 2>&1
tail -1 $O/${TAG}_bench.json | cut -c1-400; cat $O/${TAG}_steady_launch.txt $O/${TAG}_hbm_traffic.json $O/${TAG}_phase_traffic.txt
