#!/bin/bash
# randomised stress on the final tree: the whole-row GEMM (bitwise against the two-launch path), the attention kernels (after the masked-key V zeroing),
# the forward in every mode (batch invariance, determinism)
cd "$(dirname "$0")/../.."
o=gpurun_out/r04n; mkdir -p $o
python3 tools/rowln_stress.py --seconds 150 --seed 1 > $o/rowln_stress.txt 2>&1; tail -2 $o/rowln_stress.txt
python3 tools/attn_stress.py --seconds 120 --seed 4 > $o/attn_stress.txt 2>&1; tail -2 $o/attn_stress.txt
python3 tools/forward_stress.py --seconds 150 --seed 5 > $o/forward_stress.txt 2>&1; tail -2 $o/forward_stress.txt
