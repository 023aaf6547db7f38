#!/bin/bash
# the GPU suite + smoke on the final tree (record for profiles/r04_pytest_gpu.txt)
mkdir -p gpurun_out/r04t
timeout 1500 python3 -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/r04t/pytest_gpu.txt
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r04t/smoke.txt 2>&1
timeout 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -s -k bench_sizes 2>&1 | grep "^\[" > gpurun_out/r04t/fullsize_errors.txt
cat gpurun_out/r04t/pytest_gpu.txt; tail -2 gpurun_out/r04t/smoke.txt
