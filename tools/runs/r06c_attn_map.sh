#!/bin/bash
# XCD-strided block walk of the pipelined attention kernel against the round-3 walk, same box, interleaved; attention kernel tests
O=gpurun_out/r06c; mkdir -p $O
timeout 900 python3 tools/attn_map_ab.py > $O/attn_map_ab.txt 2>&1
grep -v amdgpu.ids $O/attn_map_ab.txt
timeout 900 python3 -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k attention 2>&1 | tail -3 | tee $O/pytest_attention.txt
