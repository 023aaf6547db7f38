#!/bin/bash
# CU-partition probe with micro-batches of 32 pairs (two of them = the north-star operating point B = 64), finer splits
O=gpurun_out/r06b; mkdir -p $O
timeout 900 python3 tools/cu_partition.py --pairs 32 --splits 26 24 22 --rounds 3 > $O/cu_partition_b32.txt 2>&1
grep -v amdgpu.ids $O/cu_partition_b32.txt
