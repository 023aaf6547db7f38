#!/bin/bash
# A/B: the order of the MFMAs inside a cluster of the 3-term GEMM (the compiler's order: W fragment changes with every MFMA; VTQ_MMA_ORDER=1:
# W fragment held over four MFMAs, serpentine over the row blocks, pinned with opaque accumulators in =2).  Same bits.  Interleaved on one box.
cd "$(dirname "$0")/../.."
o=gpurun_out/r04v; mkdir -p $o
bash tools/build_abl.sh mmaorder1 "-DVTQ_MMA_ORDER=1" > $o/b1.txt 2>&1 || { tail -5 $o/b1.txt; exit 1; }
bash tools/build_abl.sh mmaorder2 "-DVTQ_MMA_ORDER=2" > $o/b2.txt 2>&1 || { tail -5 $o/b2.txt; exit 1; }
for r in 1 2 3; do
  for v in shipped mmaorder1 mmaorder2; do
    if [ $v = shipped ]; then unset VTQ_LIB_PATH; else export VTQ_LIB_PATH=tools/_abl/$v.so; fi
    echo "## $v (round $r)" | tee -a $o/gemm.txt
    timeout 300 python3 tools/gemm_bench.py --fmt fp16x3 --rounds 5 2>&1 | grep -v amdgpu.ids | tee -a $o/gemm.txt
  done
done
unset VTQ_LIB_PATH
for r in 1 2 3; do
  for v in shipped mmaorder1; do
    if [ $v = shipped ]; then unset VTQ_LIB_PATH; else export VTQ_LIB_PATH=tools/_abl/$v.so; fi
    echo "## $v (round $r)" | tee -a $o/bench.txt
    VTQ_ALLOW_ABI_MISMATCH=0 timeout 300 python3 bench.py --no-cpu-baseline --no-fidelity --no-secondary --no-e2e --no-north-star --no-live-traffic --no-collective-check --no-second-mode 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])" | tee -a $o/bench.txt
  done
done
