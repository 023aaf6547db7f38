#!/bin/bash
# LayerNorm inside the residual GEMMs, in the engine: A/B of the forward (same box, interleaved), class profiles, parity tests
cd "$(dirname "$0")/../.."
o=gpurun_out/r04g; mkdir -p $o
for r in 1 2 3; do
  python3 tools/class_profile.py --precision fp16x3 2>&1 | grep -v amdgpu.ids >> $o/class_profile_ab.txt
  python3 tools/class_profile.py --precision fp16x3 --fused-ln 2>&1 | grep -v amdgpu.ids >> $o/class_profile_ab.txt
done
python3 tools/class_profile.py --precision fp16x3 --batch 64 2>&1 | grep -v amdgpu.ids >> $o/class_profile_ab.txt
python3 tools/class_profile.py --precision fp16x3 --batch 64 --fused-ln 2>&1 | grep -v amdgpu.ids >> $o/class_profile_ab.txt
cat $o/class_profile_ab.txt
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -q -x > $o/pytest_parity.txt 2>&1; tail -4 $o/pytest_parity.txt
