#!/bin/bash
# GPU box: full -m gpu suite (all failures listed), then GEMM column-group sweep on fc1 / qkv
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02e; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -25 $O/pytest_gpu.txt
for cg in 1 2 3 4 6 12; do
  VTQ_GEMM_CG=$cg timeout 300 python tools/gemm_bench.py --fmt fp16x3 fp16 --only fc1 qkv --rounds 9 2>&1 | grep -v amdgpu | sed "s/^/cg=$cg /" >> $O/cg_sweep.txt
done
cat $O/cg_sweep.txt
