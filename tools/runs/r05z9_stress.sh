#!/bin/bash
# randomised bit-for-bit stress on the final tree of round 5 (balanced GEMM epilogue): GEMM (every tile shape forced against the rule's choice: the small-tile kernels
# keep their own epilogue, so this compares the new staged form with an independent one bit for bit), whole forward, parity fuzz
cd "$(dirname "$0")/../.."
o=gpurun_out/r05z9; mkdir -p $o
timeout 400 python3 tools/gemm_stress.py --seconds 300 --seed 51 > $o/gemm_stress.txt 2>&1; tail -1 $o/gemm_stress.txt
timeout 400 python3 tools/forward_stress.py --seconds 240 --seed 52 > $o/forward_stress.txt 2>&1; tail -1 $o/forward_stress.txt
timeout 600 python3 tools/fuzz_parity.py --seed 53 > $o/fuzz.txt 2>&1; tail -3 $o/fuzz.txt
