#!/bin/bash
# GPU box: full -m gpu suite, golden error table (with the trained-like goldens), bench line (sustained + fidelity), head eager vs
# hipGraph, soak with timing
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03e; mkdir -p $O
timeout 1800 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -8 $O/pytest_gpu.txt
timeout 300 python3 tools/head_bench.py > $O/head_graph.txt 2>&1; grep -v amdgpu $O/head_graph.txt
timeout 900 python3 bench.py > $O/bench.json 2> $O/bench.err; tail -c 6000 $O/bench.json; tail -5 $O/bench.err
timeout 600 python3 tools/golden_errors.py > $O/golden_errors.txt 2>&1; tail -30 $O/golden_errors.txt
timeout 600 python3 tools/soak.py --reps 200 --modes fp16x3 fp16 > $O/soak.txt 2>&1; grep -v amdgpu $O/soak.txt
