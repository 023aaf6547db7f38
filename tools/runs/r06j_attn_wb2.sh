#!/bin/bash
# write_block with both planes staged in ONE LDS round trip (lo plane in the free ring slot's pieces of the wave) against the committed tree (wbprev), interleaved
cd "$(dirname "$0")/../.."
o=gpurun_out/r06j; mkdir -p $o
timeout 900 python3 -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "attention" 2>&1 | tail -3 | tee $o/pytest_attention.txt
for r in 1 2 3; do
  for v in shipped wbprev; do
    if [ $v = shipped ]; then unset VTQ_LIB_PATH; else export VTQ_LIB_PATH=tools/_abl/$v.so; fi
    echo "## $v (round $r)" | tee -a $o/classes.txt
    timeout 300 python3 tools/class_profile.py --precision fp16x3 --steps 20 2>&1 | grep -E "ms/step unprofiled|attention" | tee -a $o/classes.txt
  done
done
unset VTQ_LIB_PATH
timeout 200 python3 tools/attn_stress.py --seconds 90 --seed 71 2>&1 | tail -1 | tee $o/attn_stress.txt
