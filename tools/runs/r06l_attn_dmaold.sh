#!/bin/bash
# pipelined attention kernel: the LDS-DMA of a tile issued by waves 0 - 3 only (the older wave of every SIMD issues its partner's eighth too; -DVTQ_SW_DMA_OLD=1)
# against the shipped form (every wave its own eighth), one box, interleaved; attention tests on the variant
cd "$(dirname "$0")/../.."
o=gpurun_out/r06l; mkdir -p $o
VTQ_LIB_PATH=tools/_abl/dmaold.so timeout 900 python3 -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "attention" 2>&1 | tail -3 | tee $o/pytest_attention.txt
for r in 1 2 3; do
  for v in shipped dmaold; do
    if [ $v = shipped ]; then unset VTQ_LIB_PATH; else export VTQ_LIB_PATH=tools/_abl/$v.so; fi
    echo "## $v (round $r)" | tee -a $o/classes.txt
    timeout 300 python3 tools/class_profile.py --precision fp16x3 --steps 20 2>&1 | grep -E "ms/step unprofiled|attention" | tee -a $o/classes.txt
  done
done
unset VTQ_LIB_PATH
for v in shipped dmaold; do
  if [ $v = shipped ]; then unset VTQ_LIB_PATH; else export VTQ_LIB_PATH=tools/_abl/$v.so; fi
  timeout 300 python3 tools/attn_map_ab.py --shapes 8x2501x768 8x5001x768 --rounds 1 2>&1 | grep -v amdgpu | sed "s/^/$v /" | tee -a $o/shapes.txt
done
