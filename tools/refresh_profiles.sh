#!/bin/bash
# Regenerates every artefact under profiles/ on a GPU box (run through gpurun from the repo root):
#   tools/refresh_profiles.sh <tag>      e.g. r6   -> gpurun_out/profiles_<tag>/
# For each of the three measured grids (BASELINE configs[1] 256x128, configs[3] 512x256, reference-native 384x192): the
# bench line with cpu_baseline and roofline.traffic (HBM bytes per launch from two PMC passes on the same sources), and the
# lock-step phase times.  For the headline grid also: rocprofv3 kernel stats, steady launch average, SQ / LDS counters,
# per-phase counters and traffic, per-level profile, the perf-mode line with its counters, the march micro-benchmark and
# the memory-placement evidence (tools/ab_inproc.py).
TAG=${1:-r6}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/profiles_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
S="--no-cpu-baseline --steps 8 --warmup 2 --repeats 1"
# ---- HBM traffic per launch, every grid (separate passes: TCC has 4 counter slots) ----
for g in "16 100 " "32 400 _cfg4" "24 500 _native"; do
  set -- $g; res=$1; re=$2; sfx=$3
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_f$sfx -- $B $S --res $res --reynolds $re > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_w$sfx -- $B $S --res $res --reynolds $re > /dev/null 2>&1
  (cd $R; python3 tools/pmc_traffic.py $(find $O/pmc_f$sfx -name "*counter_collection.csv" | head -1) $(find $O/pmc_w$sfx -name "*counter_collection.csv" | head -1) $O/${TAG}_hbm_traffic$sfx.json "${res}" > /dev/null
   cp $O/${TAG}_hbm_traffic$sfx.json $R/profiles/${TAG}_hbm_traffic$sfx.json)   # bench.py quotes it as roofline.traffic (same kernel sources only)
done
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $O/pmc_sq -- $B $S > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR --output-format csv -d $O/pmc_sq2 -- $B $S > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU --output-format csv -d $O/pmc_perf -- $B $S --gs-mode 1 > /dev/null 2>&1
for g in "16 " "32 _cfg4" "24 _native"; do
  set -- $g
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pt_f$2 -- python3 $R/tools/phase_traffic.py run $1 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pt_w$2 -- python3 $R/tools/phase_traffic.py run $1 > /dev/null 2>&1
done
cd $R
# ---- bench lines (the driver's command for the headline; >= 64 agent steps x 3 passes and cpu_baseline for every grid) ----
timeout 900 python bench.py > $O/${TAG}_bench.json 2> $O/bench.err
timeout 1200 python bench.py --res 32 --reynolds 400 > $O/${TAG}_bench_cfg4.json 2>> $O/bench.err
timeout 900 python bench.py --res 24 --reynolds 500 > $O/${TAG}_bench_native.json 2>> $O/bench.err
timeout 600 python bench.py --no-cpu-baseline --gs-mode 1 > $O/${TAG}_bench_perfmode.json 2>> $O/bench.err
(cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B --no-cpu-baseline --steps 16 --repeats 2 > $O/${TAG}_bench_under_rocprof.json 2>/dev/null)
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/${TAG}_bench_kernel_stats.csv
python3 tools/steady_kernel_avg.py $(find $O/stats -name "*kernel_trace.csv" | head -1) > $O/${TAG}_steady_launch.txt
python3 tools/pmc_summary.py $O/pmc_sq > $O/${TAG}_sq_counters.json
python3 tools/pmc_summary.py $O/pmc_sq2 > $O/${TAG}_sq_counters2.json
python3 tools/pmc_summary.py $O/pmc_perf > $O/${TAG}_sq_counters_perfmode.json
python3 tools/phase_traffic.py report $O/pt_f $O/pt_w 16 > $O/${TAG}_phase_traffic.txt
python3 tools/phase_traffic.py report $O/pt_f_cfg4 $O/pt_w_cfg4 32 > $O/${TAG}_phase_traffic_cfg4.txt
python3 tools/phase_traffic.py report $O/pt_f_native $O/pt_w_native 24 > $O/${TAG}_phase_traffic_native.txt
for g in "16 " "32 _cfg4" "24 _native"; do
  set -- $g
  timeout 400 python tools/phase_times.py 1024 $1 2>&1 | grep -v amdgpu.ids > $O/${TAG}_phase_times_E1024$2.txt
done
UPLEG=1 timeout 300 python tools/phase_times.py 1 16 2>&1 | grep -v amdgpu.ids > $O/${TAG}_phase_times_E1.txt
timeout 300 python tools/level_profile.py 1024 8 2>&1 | grep -v amdgpu.ids > $O/${TAG}_level_profile_E1024.txt
UPLEG=1 timeout 300 python tools/level_profile.py 1 8 2>&1 | grep -v amdgpu.ids > $O/${TAG}_level_profile_E1.txt
rm -rf $O/stats $O/pmc_f* $O/pmc_w* $O/pmc_sq $O/pmc_sq2 $O/pmc_perf $O/pt_f* $O/pt_w*
bash tools/phase_counters.sh $O/phase_counters > /dev/null 2>&1
cat $O/phase_counters/p1.txt $O/phase_counters/p2.txt $O/phase_counters/p3.txt > $O/${TAG}_phase_counters.txt
rm -rf $O/phase_counters
# ---- memory placement: the same library, env instances with and without the placement search, launches interleaved
# (two processes: instances of one process compete for the fast placements) ----
L=rlfluidcontrol_amd/libafc_env.so
(echo "AFC_PLACEMENT_TRIALS=1 (first allocation taken), six instances of one process:"; python tools/ab_inproc.py --blocks 6 --per-block 6 --action-scale 0 $L:AFC_PLACEMENT_TRIALS=1 $L:AFC_PLACEMENT_TRIALS=1 $L:AFC_PLACEMENT_TRIALS=1 $L:AFC_PLACEMENT_TRIALS=1 $L:AFC_PLACEMENT_TRIALS=1 $L:AFC_PLACEMENT_TRIALS=1 2>&1 | grep "ms per launch";
 echo "default (best of 16 to 32 candidate allocations by the streaming probe), three instances of one process:"; AFC_PRINT_ALLOC=1 python tools/ab_inproc.py --blocks 6 --per-block 6 --action-scale 0 $L $L $L 2>&1 | grep "ms per launch\|candidate") > $O/${TAG}_placement.txt
(cd tools; for kp in 36 42; do for nw in 1 2 4; do /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -DAFC_ONLY_MODE=0 -DNW=$nw -DAFC_KPAD_FORCE=$kp march_bench.hip -o /tmp/mb 2>/dev/null && echo "pitch pad $kp, marching waves per workgroup $nw" && /tmp/mb 0 | grep -v "workgroups=   1"; done; done) > $O/${TAG}_march_bench.txt 2>&1
# ---- launch time against the number of envs (1 env .. 4 workgroups per CU): what ships (afc_config.upleg = 0: the march
# down the rows below 2.5 envs per CU, the strip pipeline above) and the two up-legs forced, on the SAME memory when a
# -DAFC_DEVHOOKS=1 build of these sources is there (tools/build_variant.sh r6, built before the gpurun call) ----
V=tools/bin/libafc_r6.so
if [ -f $V ]; then SH="--share-slab"; else V=$L; SH=""; fi
(for E in 1 64 256 512 640 768 1024; do python tools/ab_inproc.py $SH --envs $E --blocks 4 --per-block 4 $V:upleg=1 $V:upleg=3 $V 2>&1 | grep "ms per launch" | sed "s/^/E=$E  /"; done) > $O/${TAG}_launch_time_by_envs.txt
# ---- base priorities of the CU mates (round 6: rotating in the one-tile kernels, static in the 192-column ones): this build against the same sources with -DAFC_PRIO_ROT=-1
# (tools/bin/libafc_r5code.so: round 5's statements) on the same memory, and the mates' timeline inside a launch of the shipped
# instantiation with and without them (-DAFC_EXP_STAMPS=1 builds: tools/bin/libafc_stamp0.so / libafc_stamp1.so) ----
W=tools/bin/libafc_r5code.so
if [ -f $V ] && [ -f $W ]; then
  (for E in 1024 768 512 256; do echo "## 256x128, $E envs"; python tools/ab_inproc.py --share-slab --envs $E --blocks 8 --per-block 4 $W $V 2>&1 | grep "ms per launch"; done
   echo "## 128x64, 1024 envs"; python tools/ab_inproc.py --share-slab --res 8 --blocks 8 --per-block 4 $W $V 2>&1 | grep "ms per launch"
   echo "## 512x256 Re 400, 1024 envs (its kernels keep 0 for all)"; python tools/ab_inproc.py --share-slab --res 32 --reynolds 400 --blocks 5 --per-block 3 $W $V 2>&1 | grep "ms per launch"
   echo "## 384x192 Re 500, 1024 envs (its kernels: static 0 / 0 / 1 / 3 by mate)"; python tools/ab_inproc.py --share-slab --res 24 --reynolds 500 --blocks 5 --per-block 3 $W $V 2>&1 | grep "ms per launch") > $O/${TAG}_priority_rotation_ab.txt
fi
if [ -f tools/bin/libafc_stamp0.so ]; then
  (for v in stamp0 stamp1; do echo "== the shipped instantiation, $v (0: all base priorities 0 as in round 5; 1: rotating)"; AFC_ENV_LIB=tools/bin/libafc_$v.so python tools/desync_probe.py 0 2>&1 | grep -v amdgpu.ids | tail -9; done
   echo "== the diagnostic instantiation (rotating)"; python tools/desync_probe.py 0 2>&1 | grep -v amdgpu.ids | tail -5) > $O/${TAG}_cu_mates_timeline.txt
fi
# ---- one level's up-leg in isolation: barrier form | strip pipeline | march down the rows, bits and time; rows that cross a body ----
(/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-unused-value -Wno-unused-variable -Wno-unused-function -Iinclude -DAFC_TM_PROF=1 -o /tmp/ub tools/upleg_bench.hip 2>/dev/null && for a in "128 0 1 1 1" "64 1 0 0 1" "32 1 0 0 1"; do /tmp/ub $a; done) > $O/${TAG}_upleg_bench.txt 2>&1
# ---- bench lines of a small batch (256 envs = one per CU): what ships there (the march down the rows) and the strip pipeline ----
timeout 300 python bench.py --no-cpu-baseline --envs 256 > $O/${TAG}_bench_256envs.json 2>> $O/bench.err
timeout 300 python bench.py --no-cpu-baseline --envs 256 --upleg 1 > $O/${TAG}_bench_256envs_strip_pipeline.json 2>> $O/bench.err
UPLEG=3 timeout 300 python tools/phase_times.py 1 16 2>&1 | grep -v amdgpu.ids > $O/${TAG}_phase_times_E1_rows_march.txt
UPLEG=3 timeout 300 python tools/level_profile.py 1 8 2>&1 | grep -v amdgpu.ids > $O/${TAG}_level_profile_E1_rows_march.txt
(/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-value -o /tmp/vp tools/vmm_probe.hip 2>/dev/null && /tmp/vp 1740 10) > $O/${TAG}_vmm_placement_probe.txt 2>&1
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do python bench.py --no-cpu-baseline --steps 8 --repeats 2 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['copy_ceiling_measured_GBs'])"; done > $O/${TAG}_placement_reliability.txt
tail -1 $O/${TAG}_bench.json | cut -c1-400; cat $O/${TAG}_steady_launch.txt $O/${TAG}_hbm_traffic.json $O/${TAG}_phase_traffic.txt
