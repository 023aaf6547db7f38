#!/bin/bash
# GPU box: fp8 mode with calibrated activation scales: its tests, then the mode-fidelity table (flat-init and trained-like weights; fp8 calibrated and static)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03f; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_fp8.py -x -q -s > $O/pytest_fp8.txt 2>&1; tail -15 $O/pytest_fp8.txt
timeout 1200 python3 tools/mode_fidelity.py --pairs 256 > $O/mode_fidelity.txt 2>&1; grep -v amdgpu $O/mode_fidelity.txt | tail -30
