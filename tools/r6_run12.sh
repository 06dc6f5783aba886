#!/bin/bash
mkdir -p gpurun_out/r6
cd /root/repo
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r6/pytest_gpu.log 2>&1
echo "pytest rc $?" >> gpurun_out/r6/pytest_gpu.log
tail -8 gpurun_out/r6/pytest_gpu.log
rm -f gpurun_out/r6/ab_r6.txt
for E in 1024 640 512 256 64 1; do
  echo "== envs $E: r5 | r6 (auto)" >> gpurun_out/r6/ab_r6.txt
  timeout 400 python3 tools/ab_inproc.py --share-slab --envs $E --blocks 4 --per-block 4 tools/bin/libafc_nt0.so tools/bin/libafc_r6.so 2>&1 | grep "libafc" >> gpurun_out/r6/ab_r6.txt
done
cat gpurun_out/r6/ab_r6.txt
