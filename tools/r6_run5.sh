#!/bin/bash
mkdir -p gpurun_out/r6
cd /root/repo
rm -f gpurun_out/r6/solo.txt
for b in upleg_bench_prof ub_NOAPPLY ub_SLEEP8; do
 for args in "128 0 1 1 0 1" "128 0 1 1 0 1024"; do
  echo "== $b $args" >> gpurun_out/r6/solo.txt
  timeout 120 tools/bin/$b $args 2>&1 | grep "prof\|mg_up_tr " >> gpurun_out/r6/solo.txt
 done
done
cat gpurun_out/r6/solo.txt
