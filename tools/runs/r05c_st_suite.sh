#!/bin/bash
# product build with the tile rule: kernel + parity tests that touch the GEMMs, then the small-batch table AFTER
O=gpurun_out/r05c; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "gemm" 2>&1 | tail -5 > $O/pytest_gemm.txt
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -5 > $O/pytest_parity.txt
timeout 600 python3 tools/small_batch.py --classes --json $O/after.json > $O/after.txt 2>&1
timeout 300 python3 tools/small_batch.py --refdefault --batches 1 16 --patches 512 --classes --json $O/after_refdefault.json > $O/after_refdefault.txt 2>&1
cat $O/pytest_gemm.txt $O/pytest_parity.txt; grep -v "^ " $O/after.txt; grep -v "^ " $O/after_refdefault.txt
