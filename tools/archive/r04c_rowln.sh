#!/bin/bash
# first light of the whole-row residual GEMM with LayerNorm in its epilogue: kernel tests (bitwise against the two-launch path) + same-box A/B
cd "$(dirname "$0")/../.."
o=gpurun_out/r04c; mkdir -p $o
timeout 900 python3 -m pytest tests/test_gpu_kernels.py -q -x -k "rowln" > $o/pytest_rowln.txt 2>&1
tail -15 $o/pytest_rowln.txt
timeout 600 python3 tools/rowln_bench.py > $o/rowln_bench.txt 2>&1; grep -v amdgpu.ids $o/rowln_bench.txt
