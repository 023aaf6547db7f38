#!/bin/bash
cd "$(dirname "$0")/../.."
o=gpurun_out/r03zk; mkdir -p $o
for i in 1 2 3; do
for st in 0 2 3; do
  VTQ_GEMM_STAGGER=$st python3 tools/gemm_bench.py --fmt fp16x3 --only fc1 --rounds 5 2>&1 | grep -v amdgpu | sed "s/^/stagger=$st /" >> $o/gemm.txt
done; done
cat $o/gemm.txt | cut -c1-150
for i in 1 2; do
for st in 0 2 3; do
VTQ_GEMM_STAGGER=$st python3 bench.py --no-cpu-baseline --no-fidelity --no-second-mode --no-north-star --no-sustained --no-live-traffic 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('stagger=$st: %.1f pairs/s  %.3f ms/step  fc1 %.1f us' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'] * 1e3))" >> $o/bench_ab.txt
done; done
cat $o/bench_ab.txt
