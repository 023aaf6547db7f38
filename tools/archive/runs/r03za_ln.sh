#!/bin/bash
# LayerNorm with two rows per wave (VTQ_LN_RPW=2) against one: tests on the variant, class profile interleaved
cd "$(dirname "$0")/../.."
o=gpurun_out/r03za; mkdir -p $o
VTQ_LIB_PATH=$PWD/tools/_abl/lnrpw2.so python3 -m pytest tests/test_gpu_kernels.py -q -x -k "layernorm or ln" > $o/pytest.txt 2>&1; tail -1 $o/pytest.txt
for i in 1 2 3; do
for n in rpw1 rpw2; do
  lib=""; [ $n = rpw2 ] && lib=$PWD/tools/_abl/lnrpw2.so
  VTQ_LIB_PATH=$lib python3 tools/class_profile.py 2>&1 | grep -v amdgpu | grep "B=32\|layernorm" | sed "s/^/$n: /" >> $o/class_ab.txt
done; done
cat $o/class_ab.txt
