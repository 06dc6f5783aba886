#!/bin/bash
mkdir -p gpurun_out/r6
cd /root/repo
rm -f gpurun_out/r6/ab_tr.txt
for E in 1024 768 512 256 1; do
  echo "== envs $E" >> gpurun_out/r6/ab_tr.txt
  timeout 300 python3 tools/ab_inproc.py --share-slab --envs $E --blocks 4 --per-block 4 tools/bin/libafc_nt0.so tools/bin/libafc_tr.so 2>&1 | grep "libafc\|env 0" >> gpurun_out/r6/ab_tr.txt
done
cat gpurun_out/r6/ab_tr.txt
