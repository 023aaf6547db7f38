#!/bin/bash
# A/B: residual epilogue (out-proj, fc2) with the copy-out's LDS reads issued before the next chunk's conversion (-DVTQ_RESID_DEFER=1) against shipped.  Same bits.
cd "$(dirname "$0")/../.."
o=gpurun_out/r05z5; mkdir -p $o
export VTQ_LIB_PATH=tools/_abl/residdefer.so
timeout 900 python3 -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "gemm" 2>&1 | tail -3 | tee $o/pytest_gemm.txt
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "golden and not bench_sizes" 2>&1 | tail -3 | tee $o/pytest_parity.txt
unset VTQ_LIB_PATH
for r in 1 2 3; do
  for v in shipped residdefer; do
    if [ $v = shipped ]; then unset VTQ_LIB_PATH; else export VTQ_LIB_PATH=tools/_abl/$v.so; fi
    echo "## $v (round $r)" | tee -a $o/classes.txt
    timeout 300 python3 tools/class_profile.py --precision fp16x3 --steps 20 2>&1 | grep -E "ms/step|out_proj|fc2|fc1|qkv" | tee -a $o/classes.txt
  done
done
unset VTQ_LIB_PATH
