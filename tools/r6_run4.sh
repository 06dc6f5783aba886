#!/bin/bash
bash tools/r6_run2.sh > /dev/null 2>&1
bash tools/r6_run3.sh > /dev/null 2>&1
grep "mg_up_tr \|==\|differs" gpurun_out/r6/upleg2.txt | head -80
grep "==\|prof" gpurun_out/r6/upleg3.txt
bash tools/r6_run8.sh 2>&1 | grep "k<"
