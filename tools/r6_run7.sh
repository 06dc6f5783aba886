#!/bin/bash
mkdir -p gpurun_out/r6
cd /root/repo
export AFC_PLACEMENT_TRIALS=1
for u in 0 1; do
  for E in 1024 1; do
    echo "== UPLEG=$u E=$E" >> gpurun_out/r6/phases.txt
    UPLEG=$u timeout 300 python3 tools/phase_times.py $E 16 >> gpurun_out/r6/phases.txt 2>&1
  done
done
cat gpurun_out/r6/phases.txt
