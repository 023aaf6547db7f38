#!/bin/bash
# GPU box: full -m gpu suite, golden error table, bench line, GEMM A/B (persistent lists vs hardware dispatch, no epilogue)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02c; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -15 $O/pytest_gpu.txt
timeout 600 python tools/golden_errors.py > $O/golden_errors.txt 2>&1; tail -45 $O/golden_errors.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 3000 $O/bench.json
for fl in 0 4 8 12 0 4; do
  VTQ_GEMM_FLAGS=$fl timeout 300 python tools/gemm_bench.py --fmt fp16x2 fp16x3 --rounds 9 >> $O/gemm_ab.txt 2>&1
done
cat $O/gemm_ab.txt
timeout 200 python tools/attn_bench.py > $O/attn.txt 2>&1; cat $O/attn.txt
timeout 300 python tools/class_profile.py > $O/class_profile.txt 2>&1; cat $O/class_profile.txt
