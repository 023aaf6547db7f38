#!/bin/bash
# A/B on the balanced epilogue: GELU polynomial scalar (shipped) against packed (-DVTQ_GELU_PACKED=1).  Same bits.
cd "$(dirname "$0")/../.."
o=gpurun_out/r05z3; mkdir -p $o
for r in 1 2 3 4; do
  for v in shipped balpk; do
    if [ $v = shipped ]; then unset VTQ_LIB_PATH; else export VTQ_LIB_PATH=tools/_abl/$v.so; fi
    echo "## $v (round $r)" | tee -a $o/bench.txt
    timeout 300 python3 bench.py --no-cpu-baseline --no-fidelity --no-secondary --no-e2e --no-north-star --no-live-traffic --no-collective-check --no-second-mode --no-latency --no-practical-peak --no-auto-overhead 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])" | tee -a $o/bench.txt
  done
done
unset VTQ_LIB_PATH
