#!/bin/bash
mkdir -p gpurun_out/r6
cd /root/repo
rm -f gpurun_out/r6/upleg2.txt
for args in "128 0 1 1 0" "128 0 1 1 1" "64 1 0 0 0" "64 1 0 0 1" "32 1 0 0 0" "32 1 0 0 1" "128 0 1 0 1" "64 0 1 1 1" "32 0 1 1 1"; do
  echo "== upleg_bench $args" >> gpurun_out/r6/upleg2.txt
  timeout 120 tools/bin/upleg_bench $args >> gpurun_out/r6/upleg2.txt 2>&1
  echo "rc $?" >> gpurun_out/r6/upleg2.txt
done
grep -v "x differs" gpurun_out/r6/upleg2.txt | grep "mg_up_tr\|==\|rc"
grep "x differs" gpurun_out/r6/upleg2.txt | head -20
