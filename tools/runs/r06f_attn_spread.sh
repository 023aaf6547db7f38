#!/bin/bash
# Pipelined attention kernel: seam work spread over the MFMA groups (VTQ_SW_SPREAD) -- tests, then shipped vs spread0 (bursts, with Q prefetch + early write)
# vs seam0 (round 5), one box, interleaved; stamps of the shipped form
cd "$(dirname "$0")/../.."
o=gpurun_out/r06f; mkdir -p $o
timeout 900 python3 -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "attention" 2>&1 | tail -3 | tee $o/pytest_attention.txt
for r in 1 2 3; do
  for v in shipped spread0 seam0; do
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
VTQ_LIB_PATH=tools/_abl/adiag.so timeout 200 python3 tools/attn_probe.py --variant 1 --fmt fp16x3 --tag adiag 2>&1 | grep -v amdgpu | tee -a $o/stamps.txt
