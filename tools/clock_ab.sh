#!/bin/bash
# clocks / power under the bench for two libraries (AFC_ENV_LIB override): tools/bin/libafc_base.so (build it from the
# commit to compare with: git worktree + python -m rlfluidcontrol_amd.build) against the in-tree library
for lib in tools/bin/libafc_base.so rlfluidcontrol_amd/libafc_env.so; do
  echo "== $lib"
  AFC_ENV_LIB=$lib python bench.py --no-cpu-baseline --steps 300 --repeats 3 > /tmp/b.json 2>/dev/null &
  BP=$!
  sleep 14
  for i in $(seq 12); do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | tr '\n' ' ' | sed 's/GPU\[0\]//g; s/  */ /g'
    echo
    sleep 1
  done
  wait $BP
  python3 -c "import json; d=json.loads(open('/tmp/b.json').read().strip().split(chr(10))[-1]); print(d['value'], d['ms_per_step'])"
done
