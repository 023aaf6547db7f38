#!/bin/bash
O=gpurun_out/r05h; mkdir -p $O
timeout 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "operating_point" 2>&1 | grep -E "^\[|passed|failed" > $O/operating_point_errors.txt
cat $O/operating_point_errors.txt
bash tools/runs/r05g_attn_seams.sh
