#!/bin/bash
# GPU box: split4_f16 probe; attention phase stamps (diagnostic build); A/B base (round-2 tree) vs this tree for GEMM and attention.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03d; mkdir -p $O
tools/micro/mix_probe > $O/mix_probe.txt 2>&1; cat $O/mix_probe.txt
VTQ_LIB_PATH=$PWD/tools/_abl/diag.so timeout 300 python3 tools/attn_bench.py --fmt fp16x3 fp16 2>&1 | grep -v amdgpu > $O/attn_stamps.txt; cat $O/attn_stamps.txt
for rep in 1 2 3; do
  VTQ_LIB_PATH=$PWD/tools/_abl/base.so timeout 300 python3 tools/gemm_bench.py --only fc1 qkv --fmt fp16x3 fp16 --rounds 9 2>&1 | grep flags | sed 's/^/base /' >> $O/gemm.txt
  timeout 300 python3 tools/gemm_bench.py --only fc1 qkv --fmt fp16x3 fp16 --rounds 9 2>&1 | grep flags | sed 's/^/new  /' >> $O/gemm.txt
  VTQ_LIB_PATH=$PWD/tools/_abl/base.so timeout 300 python3 tools/attn_bench.py --fmt fp16 fp16x3 2>&1 | grep attention | sed 's/^/base /' >> $O/attn.txt
  timeout 300 python3 tools/attn_bench.py --fmt fp16 fp16x3 2>&1 | grep attention | sed "s/^/new  /" >> $O/attn.txt
done
sort -k1,1 -k3,4 -s $O/gemm.txt; sort -k1,1 -k3,3 -s $O/attn.txt
timeout 900 python3 -m pytest tests/test_gpu_kernels.py -x -q 2>&1 | tail -3
