#!/bin/bash
# kernel resources + code size of the built library: tools/kres.sh [name regex]   (KEEP=file also writes the disassembly)
python3 -m rlfluidcontrol_amd.codeobj "${1:-step}"
