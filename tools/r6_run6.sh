#!/bin/bash
mkdir -p gpurun_out/r6
cd /root/repo
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r6/pytest_gpu.log 2>&1
echo "pytest rc $?" >> gpurun_out/r6/pytest_gpu.log
tail -15 gpurun_out/r6/pytest_gpu.log
timeout 300 python bench.py --steps 32 --repeats 3 > gpurun_out/r6/bench_a.json 2> gpurun_out/r6/bench_a.err
python3 -c "
import json
d=json.loads(open('gpurun_out/r6/bench_a.json').read().strip().splitlines()[-1])
print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['timing'])
"
