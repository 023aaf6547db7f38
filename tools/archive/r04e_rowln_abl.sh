#!/bin/bash
# K-loop ablations of the whole-row kernel (diagnostic builds; results wrong by design): what the LDS-DMA issue and the fragment reads cost
cd "$(dirname "$0")/../.."
o=gpurun_out/r04e; mkdir -p $o
for v in 0 1 2 3; do
  bash tools/build_abl.sh rl$v "-DVTQ_GEMM_DIAG -DVTQ_MEASURE -DVTQ_RL_ABL=$v" > $o/build$v.txt 2>&1 || { tail -5 $o/build$v.txt; exit 1; }
done
for v in 0 1 2 3; do VTQ_LIB_PATH=tools/_abl/rl$v.so python3 tools/rowln_probe.py --noln 2>&1 | grep -v amdgpu.ids | tee -a $o/rowln_abl.txt; done
