#!/bin/bash
# tools/build_variant.sh <name> [extra compiler flags ...]: builds the CURRENT sources with -DAFC_DEVHOOKS=1 (the slab hooks of
# tools/ab_inproc.py --share-slab, which the shipped library does not carry) and the extra flags into
# tools/bin/libafc_<name>.so (for tools/ab_inproc.py) without touching the in-tree library.
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/tools/bin/libafc_$name.so
tmp=$(mktemp -d /tmp/afcvar_XXXX)
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-unused-function -DAFC_DEVHOOKS=1 -I$root/include $*"
pids=()
for s in afc_env.hip step_prod.hip step_diag.hip step_perf.hip step_lat.hip geometry.cpp; do
  /opt/rocm/bin/hipcc $FLAGS -c -o $tmp/${s%.*}.o $root/rlfluidcontrol_amd/csrc/$s > $tmp/${s%.*}.log 2>&1 &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p || { grep -h "error" -A5 $tmp/*.log | head -40; exit 1; }; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $out $tmp/*.o
rm -rf $tmp
echo $out
