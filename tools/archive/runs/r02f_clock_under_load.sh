#!/bin/bash
# GPU box: effective clock / MFMA busy of the fc1 GEMM with 8 and 32 workgroups per XCD (6 tiles per workgroup in both)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02f; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cus in 8 32; do
  M=$((cus * 8 * 6 * 256 / 12))
  VTQ_GEMM_CUS=$cus rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_cus$cus -o p -- python3 $R/tools/gemm_bench.py --fmt fp16x3 fp16 --only fc1 --M $M --rounds 1 > $O/pmc_cus$cus.log 2>&1
  python3 $R/tools/summarize_prof.py pmc $O/pmc_cus$cus gemm > $O/sum_cus$cus.txt 2>&1
  cat $O/sum_cus$cus.txt
done
