#!/bin/bash
# kernel resources + code size of the built library: tools/kres.sh [filter]
L=${AFC_ENV_LIB:-rlfluidcontrol_amd/libafc_env.so}
T=$(mktemp -d)
objcopy -O binary --only-section=.hip_fatbin $L $T/fat.bin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$T/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/dev.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/dev.co | grep -E "\.name:|vgpr_count|sgpr_count|private_segment_fixed|group_segment_fixed" | paste - - - - - | sed 's/ \+/ /g' | grep -E "${1:-step}"
/opt/rocm/lib/llvm/bin/llvm-readelf -s $T/dev.co | grep -E "FUNC" | grep -E "${1:-step}" | awk '{print $3, $8}'
[ -n "$KEEP" ] && /opt/rocm/lib/llvm/bin/llvm-objdump -d $T/dev.co > $KEEP
rm -rf $T
