#!/bin/bash
# skinny stages with their epilogue operands requested up front: tests + same-box A/B against the previous kernel
cd "$(dirname "$0")/../.."
o=gpurun_out/r03z; mkdir -p $o
python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -q -x -k "skinny or head or cls or golden or repeated or diffnet" > $o/pytest.txt 2>&1; tail -1 $o/pytest.txt
for i in 1 2 3; do
for n in new base; do
  lib=""; [ $n = base ] && lib=$PWD/tools/_abl/base.so
  VTQ_LIB_PATH=$lib python3 tools/head_bench.py 2>&1 | grep -v amdgpu | sed "s/^/$n: /" >> $o/head_ab.txt
done; done
cat $o/head_ab.txt
for n in new base; do
  lib=""; [ $n = base ] && lib=$PWD/tools/_abl/base.so
  VTQ_LIB_PATH=$lib python3 tools/class_profile.py 2>&1 | grep -v amdgpu | grep -A10 "^fp16x3" | grep "fp16x3\|head" | sed "s/^/$n: /" >> $o/class_ab.txt
done
cat $o/class_ab.txt
