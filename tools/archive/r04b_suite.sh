#!/bin/bash
# full GPU suite after the fp8 flag fix + bench stdout check
cd "$(dirname "$0")/../.."
o=gpurun_out/r04b; mkdir -p $o
timeout 2400 python3 -m pytest tests -m gpu -q -s > $o/pytest_gpu.txt 2>&1
tail -8 $o/pytest_gpu.txt
grep -h "stress5_b64_n500" $o/pytest_gpu.txt | head
timeout 900 python3 bench.py --no-cpu-baseline --no-fidelity --no-north-star --no-second-mode > $o/bench.txt 2> $o/bench.err; tail -1 $o/bench.txt | cut -c1-300; wc -l $o/bench.txt
