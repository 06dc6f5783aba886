#!/bin/bash
# placement search statistics on a GPU box: N fresh bench processes, each printing the probe rate of every candidate slab,
# the one kept, the bench value and the process's wall time.   tools/place_stats.sh [N]
N=${1:-6}
for i in $(seq 1 $N); do
  t0=$(date +%s.%N)
  AFC_PRINT_ALLOC=1 python bench.py --no-cpu-baseline --steps 8 --repeats 2 > /tmp/ps_out.txt 2> /tmp/ps_err.txt
  t1=$(date +%s.%N)
  rates=$(grep "candidate" /tmp/ps_err.txt | sed -e 's/.*streams at \([0-9]*\) GB\/s.*/\1/' | tr '\n' ' ')
  kept=$(grep "kept" /tmp/ps_err.txt | sed -e 's/.*streams at \([0-9]*\) GB\/s.*/\1/')
  val=$(python3 -c "import json; d=json.loads(open('/tmp/ps_out.txt').read().strip().split(chr(10))[-1]); print(d['value'], d['ms_per_step'])")
  echo "run $i: candidates [$rates] kept $kept -> $val  (process wall $(python3 -c "print(round($t1-$t0,1))") s)"
done
