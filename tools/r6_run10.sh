#!/bin/bash
mkdir -p gpurun_out/r6
cd /root/repo
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/r6/pytest_gpu.log 2>&1
echo "pytest rc $?" >> gpurun_out/r6/pytest_gpu.log
tail -8 gpurun_out/r6/pytest_gpu.log
export AFC_PLACEMENT_TRIALS=1
rm -f gpurun_out/r6/phases.txt
for u in 0 1; do
  echo "== UPLEG=$u E=1024" >> gpurun_out/r6/phases.txt
  UPLEG=$u timeout 300 python3 tools/phase_times.py 1024 16 >> gpurun_out/r6/phases.txt 2>&1
  echo "== level profile UPLEG=$u E=1024" >> gpurun_out/r6/phases.txt
  UPLEG=$u timeout 300 python3 tools/level_profile.py 1024 8 >> gpurun_out/r6/phases.txt 2>&1
done
cat gpurun_out/r6/phases.txt
