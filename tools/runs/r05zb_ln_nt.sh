#!/bin/bash
# A/B: LayerNorm's plane stores with the streaming (nt) policy (-DVTQ_LN_NT=1) against the default write-back policy.  Same bits.
cd "$(dirname "$0")/../.."
o=gpurun_out/r05zb; mkdir -p $o
for r in 1 2 3; do
  for v in shipped lnnt; do
    if [ $v = shipped ]; then unset VTQ_LIB_PATH; else export VTQ_LIB_PATH=tools/_abl/$v.so; fi
    echo "## $v (round $r)" | tee -a $o/classes.txt
    timeout 300 python3 tools/class_profile.py --precision fp16x3 --steps 20 2>&1 | grep -E "ms/step unprofiled|layernorm|qkv|fc1 " | tee -a $o/classes.txt
  done
done
unset VTQ_LIB_PATH
