#!/bin/bash
# in-kernel stamps of the pipelined attention kernel: iterations by kind (plain / writes a block / loads Q), shipped seam handling and the round-5 one
cd "$(dirname "$0")/../.."
o=gpurun_out/r06e; mkdir -p $o
for v in adiag adiag0; do
  for shape in "64 501" "8 2501"; do
    set -- $shape
    VTQ_LIB_PATH=tools/_abl/$v.so timeout 200 python3 tools/attn_probe.py --variant 1 --fmt fp16x3 --tag $v --nseq $1 --S $2 2>&1 | grep -v amdgpu | tee -a $o/stamps.txt
  done
done
