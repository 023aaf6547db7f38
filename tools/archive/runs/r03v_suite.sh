#!/bin/bash
# full GPU suite after the attention work + sustained A/B + bench line
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
o=gpurun_out/r03v
mkdir -p $o
timeout 2400 python3 -m pytest tests -m gpu -x -q > $o/pytest_gpu.txt 2>&1
tail -5 $o/pytest_gpu.txt
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $o/smoke.txt 2>&1; tail -2 $o/smoke.txt
for v in 0 1; do python3 tools/attn_probe.py --tag shipped --variant $v >> $o/attn_probe.txt 2>&1; done
grep -v amdgpu.ids $o/attn_probe.txt
timeout 900 python3 bench.py --no-cpu-baseline --no-fidelity > $o/bench.txt 2>&1; tail -1 $o/bench.txt | cut -c1-600
