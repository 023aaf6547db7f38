#!/bin/bash
# GPU box: split4_f16 probe; A/B of the round-2 tree (tools/_abl/base.so) against this tree: GEMM epilogue with the mix-instruction
# split, attention with the mix-instruction P split and the start stagger of the second workgroup per CU.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03c; mkdir -p $O
tools/micro/mix_probe > $O/mix_probe.txt 2>&1; cat $O/mix_probe.txt
for rep in 1 2; do
  VTQ_LIB_PATH=$PWD/tools/_abl/base.so timeout 300 python3 tools/gemm_bench.py --only fc1 qkv --fmt fp16x3 --rounds 9 2>&1 | grep flags | sed 's/^/base /' >> $O/gemm.txt
  timeout 300 python3 tools/gemm_bench.py --only fc1 qkv --fmt fp16x3 --rounds 9 2>&1 | grep flags | sed 's/^/new  /' >> $O/gemm.txt
done
cat $O/gemm.txt
VTQ_LIB_PATH=$PWD/tools/_abl/base.so timeout 300 python3 tools/attn_bench.py --fmt fp16 fp16x3 2>&1 | grep attention | sed 's/^/base /' >> $O/attn.txt
for st in 0 1 2 3 4 6 8; do
  VTQ_ATTN_STAGGER=$st timeout 300 python3 tools/attn_bench.py --fmt fp16 fp16x3 2>&1 | grep attention | sed "s/^/new stagger=$st /" >> $O/attn.txt
done
VTQ_LIB_PATH=$PWD/tools/_abl/base.so timeout 300 python3 tools/attn_bench.py --fmt fp16 fp16x3 2>&1 | grep attention | sed 's/^/base /' >> $O/attn.txt
cat $O/attn.txt
timeout 900 python3 -m pytest tests/test_gpu_kernels.py -x -q 2>&1 | tail -5
