#!/bin/bash
# A/B: fp16 hi / lo split of the attention probabilities as v_cvt_pk_f16_f32 + v_fma_mixlo / mixhi_f16 (6 vector instructions per 4 values; shipped) against the
# truncating form of rounds 1 - 5a (12 per 4 values; -DVTQ_P_SPLIT_RTZ=1).  Tests on the new form first (both kernels + split form bit-identical to each other, fp64
# errors, goldens, NaN isolation), then interleaved timing on one box.
cd "$(dirname "$0")/../.."
o=gpurun_out/r05za; mkdir -p $o
timeout 900 python3 -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "attention" 2>&1 | tail -3 | tee $o/pytest_attention.txt
timeout 1200 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "golden or attention or nan or operating_point or ladder or stress" 2>&1 | tail -3 | tee $o/pytest_parity.txt
timeout 300 python3 tools/attn_ab.py --fmt fp16x3 2>&1 | grep -v amdgpu | tee $o/ab_new.txt
for r in 1 2 3; do
  for v in shipped psplitold; do
    if [ $v = shipped ]; then unset VTQ_LIB_PATH; else export VTQ_LIB_PATH=tools/_abl/$v.so; fi
    timeout 200 python3 tools/attn_probe.py --variant 1 --fmt fp16x3 --tag $v 2>&1 | grep -v amdgpu | tee -a $o/sustained.txt
    timeout 200 python3 tools/attn_probe.py --variant 0 --fmt fp16x3 --tag $v-4wave 2>&1 | grep -v amdgpu | tee -a $o/sustained.txt
  done
done
for r in 1 2 3; do
  for v in shipped psplitold; do
    if [ $v = shipped ]; then unset VTQ_LIB_PATH; else export VTQ_LIB_PATH=tools/_abl/$v.so; fi
    echo "## $v (round $r)" | tee -a $o/classes.txt
    timeout 300 python3 tools/class_profile.py --precision fp16x3 --steps 20 2>&1 | grep -E "ms/step unprofiled|attention" | tee -a $o/classes.txt
  done
done
unset VTQ_LIB_PATH
