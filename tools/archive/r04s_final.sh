#!/bin/bash
# final confirmation on the last tree: GPU suite, smoke, default bench line
mkdir -p gpurun_out/r04s
timeout 1500 python3 -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/r04s/pytest_gpu.txt
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r04s/smoke.txt 2>&1
timeout 600 python3 bench.py > gpurun_out/r04s/bench_line.json 2> gpurun_out/r04s/bench_line.err
timeout 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -s -k bench_sizes 2>&1 | grep "^\[" > gpurun_out/r04s/fullsize_errors.txt
cat gpurun_out/r04s/pytest_gpu.txt gpurun_out/r04s/smoke.txt gpurun_out/r04s/bench_line.json gpurun_out/r04s/fullsize_errors.txt
