#!/bin/bash
# (the variant lives in commit dbe3f8f; it was removed from the tree after this measurement)
# A/B: wave-private staging of the two-plane bias / GELU epilogue (no workgroup barrier inside the epilogue; -DVTQ_EPI_WAVE=1) against the shipped
# balanced passes.  Same bits.  Tests on the variant first, then interleaved timing on one box.
cd "$(dirname "$0")/../.."
o=gpurun_out/r05z4; mkdir -p $o
export VTQ_LIB_PATH=tools/_abl/epiwave.so
timeout 900 python3 -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "gemm" 2>&1 | tail -3 | tee $o/pytest_gemm.txt
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "golden or full_size or bitwise" 2>&1 | tail -3 | tee $o/pytest_parity.txt
unset VTQ_LIB_PATH
for r in 1 2 3; do
  for v in shipped epiwave; do
    if [ $v = shipped ]; then unset VTQ_LIB_PATH; else export VTQ_LIB_PATH=tools/_abl/$v.so; fi
    echo "## $v (round $r)" | tee -a $o/gemm.txt
    timeout 300 python3 tools/gemm_bench.py --fmt fp16x3 --only qkv fc1 --rounds 7 2>&1 | grep -v amdgpu.ids | tee -a $o/gemm.txt
  done
done
for r in 1 2 3; do
  for v in shipped epiwave; do
    if [ $v = shipped ]; then unset VTQ_LIB_PATH; else export VTQ_LIB_PATH=tools/_abl/$v.so; fi
    echo "## $v (round $r)" | tee -a $o/bench.txt
    timeout 300 python3 bench.py --no-cpu-baseline --no-fidelity --no-secondary --no-e2e --no-north-star --no-live-traffic --no-collective-check --no-second-mode --no-latency --no-practical-peak --no-auto-overhead 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])" | tee -a $o/bench.txt
  done
done
unset VTQ_LIB_PATH
