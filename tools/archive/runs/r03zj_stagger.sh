#!/bin/bash
# half tile first on every second workgroup (VTQ_GEMM_STAGGER=1) against the shipped order: gemm_bench, interleaved processes, then the bench step
cd "$(dirname "$0")/../.."
o=gpurun_out/r03zj; mkdir -p $o
for i in 1 2 3; do
for st in 0 1 2; do
  VTQ_GEMM_STAGGER=$st python3 tools/gemm_bench.py --fmt fp16x3 --only qkv outproj fc2 --rounds 5 2>&1 | grep -v amdgpu | sed "s/^/stagger=$st /" >> $o/gemm.txt
done; done
cat $o/gemm.txt | cut -c1-150
for i in 1 2; do
for st in 0 1 2; do
VTQ_GEMM_STAGGER=$st python3 bench.py --no-cpu-baseline --no-fidelity --no-second-mode --no-north-star --no-sustained --no-live-traffic 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('stagger=$st: %.1f pairs/s  %.3f ms/step' % (d['value'], d['ms_per_step']))" >> $o/bench_ab.txt
done; done
cat $o/bench_ab.txt
