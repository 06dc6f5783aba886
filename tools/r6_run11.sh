#!/bin/bash
mkdir -p gpurun_out/r6
cd /root/repo
rm -f gpurun_out/r6/ab_trp.txt
for E in 1024 768 512; do
  echo "== envs $E: r5 | tr all levels | pipe+tr lib with masks 0 (pipe everywhere), 6 (tr on levels 1, 2), 7 (tr everywhere)" >> gpurun_out/r6/ab_trp.txt
  timeout 400 python3 tools/ab_inproc.py --share-slab --envs $E --blocks 4 --per-block 4 tools/bin/libafc_nt0.so tools/bin/libafc_tr.so tools/bin/libafc_trp.so:AFC_TR_MASK=0 tools/bin/libafc_trp.so:AFC_TR_MASK=6 tools/bin/libafc_trp.so:AFC_TR_MASK=7 tools/bin/libafc_trp.so:AFC_TR_MASK=4 2>&1 | grep "libafc" >> gpurun_out/r6/ab_trp.txt
done
cat gpurun_out/r6/ab_trp.txt
