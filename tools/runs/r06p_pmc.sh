#!/bin/bash
# counters of the round-6 attention kernels and of the fc1 GEMM on the final tree: separate --pmc passes, --kernel-trace only beside them, the program itself after `--`
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06p; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/attn_pmc -o a -- python3 $R/tools/attn_bench.py --fmt fp16x3 fp16 > $O/attn_pmc.log 2>&1
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/attn5k_pmc -o a -- python3 $R/tools/attn_bench.py --fmt fp16x3 --nseq 8 --S 5001 > $O/attn5k_pmc.log 2>&1
G="python3 $R/tools/gemm_bench.py --only fc1 --rounds 1 --fmt fp16x3"
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/gemm_pmc -o g -- $G > $O/gemm_pmc.log 2>&1
cd $R
rm -rf $O/*/*/*kernel_trace* $O/*/*kernel_trace* 2>/dev/null
{ echo "# attention, 64 x 501 x 12 heads (fp16x3: pipelined kernel; fp16: 4-wave kernel, staged stores)"; python3 tools/summarize_prof.py pmc $O/attn_pmc attention; echo; echo "# attention, 8 x 5001 x 12 heads, fp16x3"; python3 tools/summarize_prof.py pmc $O/attn5k_pmc attention; echo; echo "# fc1 GEMM, M = 32256"; python3 tools/summarize_prof.py pmc $O/gemm_pmc gemm_pp2; } > $O/pmc_summary.txt 2>&1
cat $O/pmc_summary.txt; rm -rf $O/attn_pmc $O/attn5k_pmc $O/gemm_pmc
