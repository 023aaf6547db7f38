#!/bin/bash
# Pipelined attention kernel, seam work: L2 prefetch of the next block's Q (VTQ_SW_QPF) and the finished block's output written at the top of the
# next iteration (VTQ_SW_EARLY_WRITE) -- shipped (both) against builds without one / without both, one box, interleaved.  Tests on the shipped form first.
cd "$(dirname "$0")/../.."
o=gpurun_out/r06d; mkdir -p $o
timeout 900 python3 -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "attention" 2>&1 | tail -3 | tee $o/pytest_attention.txt
for r in 1 2 3; do
  for v in shipped seam0 qpf0 ew0; do
    if [ $v = shipped ]; then unset VTQ_LIB_PATH; else export VTQ_LIB_PATH=tools/_abl/$v.so; fi
    timeout 200 python3 tools/attn_probe.py --variant 1 --fmt fp16x3 --tag $v 2>&1 | grep -v amdgpu | tee -a $o/sustained.txt
  done
done
for r in 1 2 3; do
  for v in shipped seam0; do
    if [ $v = shipped ]; then unset VTQ_LIB_PATH; else export VTQ_LIB_PATH=tools/_abl/$v.so; fi
    echo "## $v (round $r)" | tee -a $o/classes.txt
    timeout 300 python3 tools/class_profile.py --precision fp16x3 --steps 20 2>&1 | grep -E "ms/step unprofiled|attention" | tee -a $o/classes.txt
  done
done
unset VTQ_LIB_PATH
for v in shipped seam0; do
  if [ $v = shipped ]; then unset VTQ_LIB_PATH; else export VTQ_LIB_PATH=tools/_abl/$v.so; fi
  timeout 300 python3 tools/attn_map_ab.py --shapes 64x501x768 32x1025x1024 8x2501x768 --rounds 1 2>&1 | grep -v amdgpu | sed "s/^/$v /" | tee -a $o/shapes.txt
done
unset VTQ_LIB_PATH
