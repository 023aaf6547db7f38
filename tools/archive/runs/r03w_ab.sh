#!/bin/bash
# attention variants inside the forward: same box, interleaved; class profile with the default rule
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r03w
o=gpurun_out/r03w
python3 -m pytest tests/test_gpu_kernels.py -q -k attention > $o/pytest_attention.txt 2>&1; tail -1 $o/pytest_attention.txt
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fp8.py -q -x > $o/pytest_parity.txt 2>&1; tail -1 $o/pytest_parity.txt
for v in 0 1; do python3 tools/attn_probe.py --tag shipped --variant $v 2>&1 | grep -v amdgpu >> $o/attn_probe.txt; done; cat $o/attn_probe.txt
for i in 1 2 3; do
for v in 0 1; do
VTQ_ATTN_VARIANT=$v python3 bench.py --no-cpu-baseline --no-fidelity --no-second-mode --no-north-star --no-sustained --no-live-traffic 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('variant $v: %.1f pairs/s  %.3f ms/step' % (d['value'], d['ms_per_step']))" >> $o/bench_ab.txt
done; done
cat $o/bench_ab.txt
python3 tools/class_profile.py > $o/class_profile.txt 2>&1; grep -v amdgpu $o/class_profile.txt | head -24
