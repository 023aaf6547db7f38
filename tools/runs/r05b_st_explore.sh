#!/bin/bash
# small-tile GEMM variants against the persistent 256x256 kernel (-DVTQ_GEMM_ST_EXPLORE build of the tree)
O=gpurun_out/r05b; mkdir -p $O
timeout 900 python3 tools/st_bench.py --variants 0 1 2 3 4 5 6 7 --batches 1 2 3 4 5 6 8 12 16 32 --json $O/st_explore.json > $O/st_explore.txt 2>&1
timeout 300 python3 tools/st_bench.py --variants 0 1 11 12 13 14 15 3 16 17 --batches 1 4 --only outproj fc2 > $O/st_ablation.txt 2>&1
timeout 300 python3 tools/st_bench.py --variants 0 1 2 3 --batches 1 2 4 --fmt fp16 > $O/st_fp16.txt 2>&1
timeout 300 python3 tools/st_bench.py --variants 0 1 2 3 --batches 1 2 4 --fmt fp16x2 > $O/st_fp16x2.txt 2>&1
timeout 300 python3 tools/st_bench.py --variants 0 1 2 3 --batches 1 2 4 --fmt bf16x3 > $O/st_bf16x3.txt 2>&1
timeout 300 python3 tools/st_bench.py --variants 0 1 2 3 --batches 1 2 4 --fmt bf16 > $O/st_bf16.txt 2>&1
cat $O/st_explore.txt $O/st_ablation.txt; grep -c DIFFERENT $O/st_*.txt; grep DIFFERENT $O/st_fp16.txt $O/st_fp16x2.txt $O/st_bf16x3.txt $O/st_bf16.txt | head
