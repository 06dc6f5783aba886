#!/bin/bash
# round 6, GPU call 1: the transposed up-leg against mg_up_smooth in isolation; cache-policy variants on the same memory
mkdir -p gpurun_out/r6
cd /root/repo
for args in "128 0 1 1 0" "128 0 1 1 1" "64 1 0 0 0" "64 1 0 0 1" "32 1 0 0 0" "32 1 0 0 1" "128 0 1 0 1" "64 0 1 1 1"; do
  echo "== upleg_bench $args" >> gpurun_out/r6/upleg.txt
  timeout 120 tools/bin/upleg_bench $args >> gpurun_out/r6/upleg.txt 2>&1
  echo "rc $?" >> gpurun_out/r6/upleg.txt
done
timeout 600 python3 tools/ab_inproc.py --share-slab --blocks 10 tools/bin/libafc_nt0.so tools/bin/libafc_nt1.so tools/bin/libafc_nt2.so tools/bin/libafc_nt3.so tools/bin/libafc_nt7.so tools/bin/libafc_nt31.so tools/bin/libafc_nt63.so > gpurun_out/r6/nt_ab.txt 2>&1
echo "rc $?" >> gpurun_out/r6/nt_ab.txt
tail -30 gpurun_out/r6/upleg.txt; cat gpurun_out/r6/nt_ab.txt | tail -12
