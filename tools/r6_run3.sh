#!/bin/bash
mkdir -p gpurun_out/r6
cd /root/repo
rm -f gpurun_out/r6/upleg3.txt
for args in "128 0 1 1 1 1" "128 0 1 1 1 1024" "64 1 0 0 1 1" "64 1 0 0 1 1024" "32 1 0 0 1 1" "32 1 0 0 1 1024"; do
  echo "== upleg_bench_prof $args" >> gpurun_out/r6/upleg3.txt
  timeout 120 tools/bin/upleg_bench_prof $args >> gpurun_out/r6/upleg3.txt 2>&1
  echo "rc $?" >> gpurun_out/r6/upleg3.txt
done
grep "mg_up_tr \|==\|prof" gpurun_out/r6/upleg3.txt
