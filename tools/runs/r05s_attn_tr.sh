#!/bin/bash
# 4-wave attention kernel: the V^T transposed LDS reads as asm statements (no compiler-inserted vmcnt(0) drain of the next tile's LDS-DMA in front of the PV phase)
# against the shipped kernel (builtin reads), interleaved; bitwise comparison by the attention tests on the measurement build
cd "$(dirname "$0")/../.."
O=gpurun_out/r05s; mkdir -p $O; : > $O/attn_tr.txt
for r in 1 2 3; do
  for cfg in "64 501 fp16x3" "64 501 fp16" "2 501 fp16x3" "8 501 fp16x3" "16 501 fp16x3"; do set -- $cfg
    python3 tools/attn_bench.py --nseq $1 --S $2 --fmt $3 --variant 0 2>&1 | grep attention | sed 's/^/shipped   /' >> $O/attn_tr.txt
    VTQ_LIB_PATH=tools/_abl/asmtr.so python3 tools/attn_bench.py --nseq $1 --S $2 --fmt $3 --variant 0 2>&1 | grep attention | sed 's/^/asm reads /' >> $O/attn_tr.txt
  done
done
VTQ_LIB_PATH=$PWD/tools/_abl/asmtr.so timeout 600 python3 -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k attention 2>&1 | tail -2 >> $O/attn_tr.txt
cat $O/attn_tr.txt
