#!/bin/bash
# same-box A/B of two builds of the library: tools/bin/libafc_base.so (reference) vs the in-tree one
# usage: tools/ab.sh [rounds] [extra bench args]
R=${1:-3}; shift
for i in $(seq $R); do
  a=$(AFC_ENV_LIB=tools/bin/libafc_base.so timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" 2>&1 | tail -1 | sed 's/.*"value": \([0-9.]*\).*/\1/')
  b=$(timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" 2>&1 | tail -1 | sed 's/.*"value": \([0-9.]*\).*/\1/')
  echo "base $a  new $b"
done
