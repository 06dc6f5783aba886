#!/bin/bash
# sample clocks / power while the bench runs
python bench.py --no-cpu-baseline --steps 200 --repeats 3 > /tmp/b.json 2>/dev/null &
BP=$!
sleep 12
for i in $(seq 24); do
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temperature \(Sensor (junction|memory)" | tr '\n' ' ' | sed 's/GPU\[0\]//g; s/  */ /g'
  echo
  sleep 1
done
wait $BP
tail -1 /tmp/b.json | cut -c1-200
