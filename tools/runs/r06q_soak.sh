#!/bin/bash
# more randomised evidence on the final tree: parity fuzz (two more seeds), forward stress, GEMM stress, attention stress
cd "$(dirname "$0")/../.."
O=gpurun_out/r06q; mkdir -p $O
timeout 700 python3 tools/fuzz_parity.py --seed 91 > $O/fuzz91.txt 2>&1; tail -3 $O/fuzz91.txt
timeout 700 python3 tools/fuzz_parity.py --seed 92 > $O/fuzz92.txt 2>&1; tail -3 $O/fuzz92.txt
timeout 400 python3 tools/forward_stress.py --seconds 300 --seed 93 > $O/forward_stress.txt 2>&1; tail -1 $O/forward_stress.txt
timeout 400 python3 tools/gemm_stress.py --seconds 240 --seed 94 > $O/gemm_stress.txt 2>&1; tail -1 $O/gemm_stress.txt
timeout 300 python3 tools/attn_stress.py --seconds 200 --seed 95 > $O/attn_stress.txt 2>&1; tail -1 $O/attn_stress.txt
