#!/bin/bash
# GPU box: (1) shipped GEMM: baseline, epilogue skipped, all-half-tile schedule; (2) diagnostic build: in-kernel clock (s_memtime /
# s_memrealtime) and the price of VALU work in the load phases (beside the partner wave's MFMA cluster); (3) clock vs CUs per XCD.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03a; mkdir -p $O
for fl in 0 8; do VTQ_GEMM_FLAGS=$fl timeout 300 python3 tools/gemm_bench.py --fmt fp16x3 fp16 --rounds 7 >> $O/gemm_prod.txt 2>&1; done
for fl in 0 8; do VTQ_GEMM_SCHED=2 VTQ_GEMM_FLAGS=$fl timeout 300 python3 tools/gemm_bench.py --only fc1 qkv --fmt fp16x3 fp16 --rounds 7 >> $O/gemm_halves.txt 2>&1; done
export VTQ_LIB_PATH=$PWD/tools/_abl/diag.so
for fl in 0 8; do VTQ_GEMM_FLAGS=$fl timeout 400 python3 tools/clock_probe.py --fmt fp16x3 fp16 --shadow 0 4 8 12 16 >> $O/clock_shadow.txt 2>&1; done
for cus in 8 16 32; do VTQ_GEMM_CUS=$cus timeout 300 python3 tools/clock_probe.py --M $((cus * 1024)) --shadow 0 --fmt fp16x3 fp16 >> $O/clock_cus.txt 2>&1; done
cat $O/gemm_prod.txt $O/gemm_halves.txt $O/clock_shadow.txt $O/clock_cus.txt
