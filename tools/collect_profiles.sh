#!/bin/bash
# Runs on the GPU box (gpurun): collects the rocprofv3 passes behind profiles/*.txt into gpurun_out/prof_<name>/.
# Counters are collected in their own passes (--pmc with --kernel-trace only), as the MI355X guide prescribes.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -o s -- python3 $R/bench.py --no-cpu-baseline > $O/prof_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/prof_fetch -o f -- python3 $R/tools/gemm_bench.py --only fc1 --rounds 1 > $O/prof_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/prof_write -o w -- python3 $R/tools/gemm_bench.py --only fc1 --rounds 1 > $O/prof_write.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/prof_gemm_pmc -o g -- python3 $R/tools/gemm_bench.py --only fc1 --rounds 1 > $O/prof_gemm_pmc.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/prof_attn_pmc -o a -- python3 $R/tools/attn_bench.py > $O/prof_attn_pmc.log 2>&1
cd $R
for d in stats; do python3 tools/summarize_prof.py stats $O/prof_$d > $O/sum_stats.txt 2>&1; done
python3 tools/summarize_prof.py pmc $O/prof_fetch gemm > $O/sum_fetch.txt 2>&1
python3 tools/summarize_prof.py pmc $O/prof_write gemm > $O/sum_write.txt 2>&1
python3 tools/summarize_prof.py pmc $O/prof_gemm_pmc gemm > $O/sum_gemm_pmc.txt 2>&1
python3 tools/summarize_prof.py pmc $O/prof_attn_pmc attention > $O/sum_attn_pmc.txt 2>&1
grep -h "^{\"metric\"" $O/prof_stats.log | tail -1 > $O/bench_line_profiled.json
rm -rf $O/prof_stats/*trace* 2>/dev/null
ls $O
