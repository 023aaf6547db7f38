#!/bin/bash
# round 6, first GPU call: (1) the fp8 experiment's tests through its own library handle in the default invocation, the long-sequence goldens
# (N = 5000), the pre-embedded pairwise entry; (2) the CU-partition probe (tools/cu_partition.py)
O=gpurun_out/r06a; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_fp8.py tests/test_gpu_parity.py -m gpu -q -x -k "fp8 or long_sequence or pre_embedded or token or pairwise" -s 2>&1 | grep -v "^$" | tail -40 > $O/pytest_subset.txt
tail -5 $O/pytest_subset.txt
timeout 900 python3 tools/cu_partition.py --splits 28 24 20 --rounds 3 > $O/cu_partition.txt 2>&1
cat $O/cu_partition.txt | grep -v amdgpu.ids
