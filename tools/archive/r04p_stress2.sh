#!/bin/bash
# second-seed stress runs on the final tree
cd "$(dirname "$0")/../.."
o=gpurun_out/r04q; mkdir -p $o
python3 tools/rowln_stress.py --seconds 300 --seed 2 > $o/rowln_stress.txt 2>&1; tail -1 $o/rowln_stress.txt
python3 tools/attn_stress.py --seconds 200 --seed 6 > $o/attn_stress.txt 2>&1; tail -1 $o/attn_stress.txt
python3 tools/forward_stress.py --seconds 240 --seed 7 > $o/forward_stress.txt 2>&1; tail -1 $o/forward_stress.txt
python3 tools/gemm_stress.py --seconds 120 --seed 3 > $o/gemm_stress.txt 2>&1; tail -1 $o/gemm_stress.txt
python3 tools/fuzz_parity.py --cases 250 --precision fp16x3 --seed 11 > $o/fuzz.txt 2>&1; tail -2 $o/fuzz.txt
