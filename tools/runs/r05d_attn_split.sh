#!/bin/bash
# attention at sequence lengths a few rows past a multiple of 256 (reference-default topology S = 521): 4-wave kernel, pipelined kernel, split form
O=gpurun_out/r05d; mkdir -p $O
{ for S in 521 501 1025 257 545; do for nseq in 32 64; do for v in 0 1 2; do
  timeout 120 python3 tools/attn_bench.py --fmt fp16x3 --nseq $nseq --S $S --variant $v 2>&1 | grep attention; done; done; done; } > $O/attn_split.txt
timeout 900 python3 -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "attention" 2>&1 | tail -3 > $O/pytest_attention.txt
cat $O/attn_split.txt $O/pytest_attention.txt
