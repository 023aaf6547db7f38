#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02g; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.txt 2>&1; tail -6 $O/pytest_gpu.txt
timeout 300 python tools/class_profile.py --precision fp16x3 fp16 > $O/class_profile.txt 2>&1; grep -v "amdgpu\|Warn\|warn" $O/class_profile.txt
