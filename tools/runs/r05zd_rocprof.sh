#!/bin/bash
# rocprofv3 --kernel-trace --stats of the headline bench command on the final tree of round 5 (balanced GEMM epilogue), and the line the same command printed
O=gpurun_out/r05zd; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_bench -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-second-mode --no-north-star --no-live-traffic --no-sustained --no-fidelity --no-e2e --no-secondary --no-latency --no-practical-peak --no-auto-overhead --no-collective-check > $GRAFT_REPO_ROOT/$O/bench_line_headline_profiled.json 2>/dev/null
cd $GRAFT_REPO_ROOT
python3 tools/summarize_prof.py stats $O/prof_bench > $O/bench_headline_kernel_stats.txt 2>&1; rm -rf $O/prof_bench
head -12 $O/bench_headline_kernel_stats.txt; python3 -c "
import json; d=json.load(open('$O/bench_line_headline_profiled.json')); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
