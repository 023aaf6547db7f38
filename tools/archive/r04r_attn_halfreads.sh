#!/bin/bash
# bound on what halving the K / V fragment reads per MFMA (64 query rows per wave) could buy the pipelined attention kernel: a measurement build that skips
# every second fragment group's LDS reads (results wrong by design), interleaved with the shipped kernel
cd "$(dirname "$0")/../.."
o=gpurun_out/r04r; mkdir -p $o
bash tools/build_abl.sh halfreads "-DVTQ_SW_HALFREADS=1" > $o/b.txt 2>&1 || { tail -5 $o/b.txt; exit 1; }
for r in 1 2 3; do
  python3 tools/attn_probe.py --tag shipped --variant 1 2>&1 | grep -v amdgpu.ids | grep fp16x3 | tee -a $o/halfreads.txt
  VTQ_LIB_PATH=tools/_abl/halfreads.so python3 tools/attn_probe.py --tag half_reads --variant 1 2>&1 | grep -v amdgpu.ids | grep fp16x3 | tee -a $o/halfreads.txt
done
