#!/bin/bash
# GPU suite on the product library; the fp8 experiment's tests on its own build; operating-point errors of every mode
O=gpurun_out/r05f; mkdir -p $O
timeout 2400 python3 -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/pytest_gpu.txt
VTQ_LIB_PATH=$PWD/vtamiq_amd/libvtamiq_hip_fp8.so timeout 1200 python3 -m pytest tests/test_gpu_fp8.py -m gpu -q 2>&1 | tail -5 > $O/pytest_gpu_fp8_build.txt
timeout 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "operating_point" 2>&1 | grep "^\[" > $O/operating_point_errors.txt
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
cat $O/pytest_gpu.txt $O/pytest_gpu_fp8_build.txt $O/operating_point_errors.txt; tail -2 $O/smoke.txt
