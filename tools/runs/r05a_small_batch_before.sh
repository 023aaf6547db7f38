#!/bin/bash
# round 5, first call: the small-batch regime BEFORE any change + rocprofv3 kernel stats at B = 1
O=gpurun_out/r05a; mkdir -p $O
timeout 600 python3 tools/small_batch.py --classes --json $O/before.json > $O/before.txt 2>&1
timeout 300 python3 tools/small_batch.py --refdefault --batches 1 16 --patches 512 --classes --json $O/before_refdefault.json > $O/before_refdefault.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_b1 -o p -- python3 $GRAFT_REPO_ROOT/tools/small_batch.py --batches 1 --steps 20 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/summarize_prof.py stats $O/prof_b1 > $O/b1_kernel_stats.txt 2>&1
rm -rf $O/prof_b1
cat $O/before.txt | head -80; cat $O/b1_kernel_stats.txt | head -40
