#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r6
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r6/pytest_gpu_final.log 2>&1
echo "pytest rc $?" >> gpurun_out/r6/pytest_gpu_final.log
tail -4 gpurun_out/r6/pytest_gpu_final.log
bash tools/refresh_profiles.sh r6 > gpurun_out/r6/refresh.log 2>&1
tail -3 gpurun_out/r6/refresh.log | cut -c1-300
cat gpurun_out/profiles_r6/r6_priority_rotation_ab.txt | cut -c1-200
