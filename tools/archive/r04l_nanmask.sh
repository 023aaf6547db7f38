#!/bin/bash
# masked keys' V rows zeroed in the last key tile: kernel tests, NaN isolation end to end, bitwise agreement of the two kernels, timing
cd "$(dirname "$0")/../.."
o=gpurun_out/r04l; mkdir -p $o
timeout 1500 python3 -m pytest tests/test_gpu_kernels.py -q -x -k "attention" > $o/pytest_attn.txt 2>&1; tail -3 $o/pytest_attn.txt
timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -x -k "nan or attention_kernels_agree or golden" > $o/pytest_par.txt 2>&1; tail -3 $o/pytest_par.txt
for v in 0 1; do python3 tools/attn_probe.py --tag shipped --variant $v 2>&1 | grep -v amdgpu.ids | tee -a $o/attn_probe.txt; done
