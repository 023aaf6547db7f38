#!/bin/bash
# GPU box: GEMM v2 / attention correctness + timing (persistent kernel, operand formats, ablation flags)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "split or gemm or layernorm or attention" > $O/pytest_kernels.txt 2>&1
tail -5 $O/pytest_kernels.txt
for fl in 0 2 1 3; do
  VTQ_GEMM_FLAGS=$fl timeout 600 python tools/gemm_bench.py --fmt bf16 bf16x3 fp16x2 fp16x3 --rounds 5 >> $O/gemm_bench.txt 2>&1
done
timeout 300 python tools/attn_bench.py >> $O/attn_bench.txt 2>&1
cat $O/gemm_bench.txt $O/attn_bench.txt
