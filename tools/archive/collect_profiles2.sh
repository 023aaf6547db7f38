#!/bin/bash
# Second collection pass (GPU box): (a) kernel stats of the HEADLINE-only bench run, whose fc1 average must agree with the bench
# line's roofline.avg_launch_ms; (b) L2-miss read traffic of the fc1 GEMM as a function of the tile order's column-group width.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02q; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 $R/bench.py --no-cpu-baseline --no-second-mode --no-north-star > $O/bench_line_headline_profiled.json 2> $O/stats.err
for cg in 1 3 12; do
  export VTQ_GEMM_CG=$cg
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch_cg$cg -o f -- python3 $R/tools/gemm_bench.py --only fc1 --rounds 3 --fmt fp16x3 fp16 > $O/fetch_cg$cg.log 2>&1
done
unset VTQ_GEMM_CG
cd $R
python3 tools/summarize_prof.py stats $O/stats > $O/sum_stats_headline.txt 2>&1
for cg in 1 3 12; do python3 tools/summarize_prof.py pmc $O/fetch_cg$cg gemm > $O/sum_fetch_cg$cg.txt 2>&1; grep flags $O/fetch_cg$cg.log >> $O/sum_fetch_cg$cg.txt; done
rm -rf $O/stats/*trace* $O/*/*kernel_trace* 2>/dev/null
cat $O/sum_fetch_cg*.txt; head -12 $O/sum_stats_headline.txt
