#!/bin/bash
# paired query blocks (workgroups w, w+8 share K/V tiles in one XCD's L2) against contiguous runs: same box, interleaved
cd "$(dirname "$0")/../.."
o=gpurun_out/r03x; mkdir -p $o
python3 -m pytest tests/test_gpu_kernels.py -q -k attention > $o/pytest_attention.txt 2>&1; tail -1 $o/pytest_attention.txt
for i in 1 2 3; do
for n in paired unpaired; do
  lib=""; [ $n = unpaired ] && lib=$PWD/tools/_abl/unpaired.so
  echo "$n: $(VTQ_LIB_PATH=$lib python3 tools/attn_ab.py --shapes 64x501x768 128x501x768 32x1025x1024 --fmt fp16x3 --reps 50 2>&1 | grep -v amdgpu | tr '\n' '|')" >> $o/ab.txt
done; done
cat $o/ab.txt
for i in 1 2 3; do
for n in paired unpaired; do
  lib=""; [ $n = unpaired ] && lib=$PWD/tools/_abl/unpaired.so
VTQ_LIB_PATH=$lib python3 bench.py --no-cpu-baseline --no-fidelity --no-second-mode --no-north-star --no-sustained --no-live-traffic 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$n: %.1f pairs/s  %.3f ms/step' % (d['value'], d['ms_per_step']))" >> $o/bench_ab.txt
done; done
cat $o/bench_ab.txt
