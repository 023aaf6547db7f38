#!/bin/bash
# seam de-alignment of the pipelined attention kernel's persistent grid -- every second workgroup of an XCD starts 1 / 2 / 4 / 8 us late
# (measurement builds -DVTQ_SW_SEAM_STAGGER=n), interleaved three times with the shipped kernel on one box
cd "$(dirname "$0")/../.."
o=gpurun_out/r05g; mkdir -p $o; : > $o/seams.txt
for n in 1 2 4 8; do bash tools/build_abl.sh seam$n "-DVTQ_SW_SEAM_STAGGER=$n" > $o/b$n.txt 2>&1 || { tail -5 $o/b$n.txt; exit 1; }; done
for r in 1 2 3; do
  python3 tools/attn_probe.py --tag shipped --variant 1 --fmt fp16x3 2>&1 | grep -v amdgpu.ids | grep fp16x3 | tee -a $o/seams.txt
  for n in 1 2 4 8; do
    VTQ_LIB_PATH=tools/_abl/seam$n.so python3 tools/attn_probe.py --tag stagger_${n}us --variant 1 --fmt fp16x3 2>&1 | grep -v amdgpu.ids | grep fp16x3 | tee -a $o/seams.txt
  done
done
