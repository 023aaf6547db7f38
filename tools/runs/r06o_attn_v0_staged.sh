#!/bin/bash
# 4-wave attention kernel: output staged through LDS (16-byte stores on whole row segments) against the direct 8-byte stores (v0old), one box, interleaved
cd "$(dirname "$0")/../.."
o=gpurun_out/r06o; mkdir -p $o
timeout 900 python3 -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "attention" 2>&1 | tail -3 | tee $o/pytest_attention.txt
for r in 1 2; do
  for v in shipped v0old; do
    if [ $v = shipped ]; then unset VTQ_LIB_PATH; else export VTQ_LIB_PATH=tools/_abl/$v.so; fi
    timeout 300 python3 tools/attn_map_ab.py --shapes 64x501x768 32x521x768 8x501x768 4x501x768 64x1025x768 --rounds 1 --variant 0 --warm 0.3 2>&1 | grep -v amdgpu | sed "s/^/$v fp16x3 /" | cut -c1-120 | tee -a $o/v0.txt
    timeout 300 python3 tools/attn_map_ab.py --shapes 64x501x768 32x521x768 --fmt fp16 --rounds 1 --variant 0 --warm 0.3 2>&1 | grep -v amdgpu | sed "s/^/$v fp16   /" | cut -c1-120 | tee -a $o/v0.txt
  done
done
unset VTQ_LIB_PATH
for r in 1 2; do
  for v in shipped v0old; do
    if [ $v = shipped ]; then unset VTQ_LIB_PATH; else export VTQ_LIB_PATH=tools/_abl/$v.so; fi
    echo "## $v (round $r)" | tee -a $o/classes.txt
    timeout 300 python3 tools/class_profile.py --precision fp16 --steps 20 2>&1 | grep -E "ms/step unprofiled|attention" | tee -a $o/classes.txt
    timeout 300 python3 tools/class_profile.py --precision fp16x3 --refdefault --batch 16 --patches 512 --steps 20 2>&1 | grep -E "ms/step unprofiled|attention" | tee -a $o/classes.txt
  done
done
