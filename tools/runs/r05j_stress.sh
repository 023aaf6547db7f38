#!/bin/bash
# randomised bit-for-bit stress on the round-5 tree: GEMM (every tile shape forced against the rule's choice), attention (4-wave / pipelined / split), whole forward
cd "$(dirname "$0")/../.."
o=gpurun_out/r05j; mkdir -p $o
timeout 400 python3 tools/gemm_stress.py --seconds 240 --seed 11 > $o/gemm_stress.txt 2>&1; tail -2 $o/gemm_stress.txt
timeout 400 python3 tools/gemm_stress.py --seconds 120 --seed 12 >> $o/gemm_stress.txt 2>&1; tail -1 $o/gemm_stress.txt
timeout 300 python3 tools/attn_stress.py --seconds 150 --seed 13 > $o/attn_stress.txt 2>&1; tail -2 $o/attn_stress.txt
timeout 400 python3 tools/forward_stress.py --seconds 240 --seed 14 > $o/forward_stress.txt 2>&1; tail -2 $o/forward_stress.txt
