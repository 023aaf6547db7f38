#!/bin/bash
# round 4, first call: full GPU suite on the hygiene tree (no measurement knobs in the library, engine options, RCCL on one rank,
# ladder golden), smoke, the bench line with its new blocks, and the class profile that is this round's same-day baseline
cd "$(dirname "$0")/../.."
o=gpurun_out/r04a; mkdir -p $o
timeout 2400 python3 -m pytest tests -m gpu -x -q -s > $o/pytest_gpu.txt 2>&1
tail -5 $o/pytest_gpu.txt
grep -h "stress5_b64_n500" $o/pytest_gpu.txt | head
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $o/smoke.txt 2>&1; tail -2 $o/smoke.txt
timeout 1500 python3 bench.py > $o/bench.txt 2> $o/bench.err; tail -1 $o/bench.txt | cut -c1-1500
python3 tools/class_profile.py --precision fp16x3 fp16 > $o/class_profile.txt 2>&1; cat $o/class_profile.txt | grep -v amdgpu.ids
python3 tools/class_profile.py --refdefault --batch 16 --patches 512 --precision fp16x3 > $o/class_profile_refdefault.txt 2>&1; grep -v amdgpu.ids $o/class_profile_refdefault.txt
