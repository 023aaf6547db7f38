#!/bin/bash
# GPU box: is the GEMM epilogue bound by chip-level write bandwidth?  Same tiles per workgroup (6), fewer workgroups.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02d; mkdir -p $O
for cus in 4 8 16 32; do
  M=$((cus * 8 * 6 * 256 / 12))
  for fl in 0 8; do
    VTQ_GEMM_CUS=$cus VTQ_GEMM_FLAGS=$fl timeout 300 python tools/gemm_bench.py --fmt fp16x3 fp16 --only fc1 --M $M --rounds 9 2>&1 | grep -v amdgpu | sed "s/^/cus=$cus /" >> $O/cus_sweep.txt
  done
done
cat $O/cus_sweep.txt
