#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02h; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-second-mode --no-north-star > $O/stats.log 2>&1
cd $R; python3 tools/summarize_prof.py stats $O/stats > $O/kernel_stats.txt 2>&1; cat $O/kernel_stats.txt; rm -rf $O/stats/*trace* 2>/dev/null
