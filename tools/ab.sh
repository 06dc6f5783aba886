#!/bin/bash
# same-box A/B of builds of the library: tools/ab.sh [rounds] [lib ...]   (default: tools/bin/libafc_base.so vs the in-tree one)
# each round runs every library once, alternating, so that box drift hits all of them alike
R=${1:-3}; shift
LIBS=("$@")
[ ${#LIBS[@]} -eq 0 ] && LIBS=(tools/bin/libafc_base.so rlfluidcontrol_amd/libafc_env.so)
for i in $(seq $R); do
  line=""
  if [ $((i % 2)) -eq 0 ]; then  # even rounds in reverse order: a box that slows down as it warms up hits all alike
    ORDER=(); for ((k=${#LIBS[@]}-1; k>=0; k--)); do ORDER+=("${LIBS[k]}"); done
  else
    ORDER=("${LIBS[@]}")
  fi
  for L in "${ORDER[@]}"; do
    v=$(AFC_ENV_LIB=$L timeout 300 python bench.py --steps 10 --warmup 3 --repeats 3 --no-cpu-baseline 2>&1 | tail -1 | sed 's/.*"value": \([0-9.]*\).*/\1/')
    line="$line $(basename $L .so) $v"
  done
  echo "$line"
done
