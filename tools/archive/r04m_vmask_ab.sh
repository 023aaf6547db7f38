#!/bin/bash
# A/B: the zeroing of the masked keys' V rows in the attention kernels (this tree) against the same tree without it, interleaved on one box
cd "$(dirname "$0")/../.."
o=gpurun_out/r04m; mkdir -p $o
bash tools/build_abl.sh novmask "-DVTQ_ATTN_NO_VMASK" > $o/b.txt 2>&1 || { tail -5 $o/b.txt; exit 1; }
for r in 1 2 3; do
  for v in 0 1; do
    python3 tools/attn_probe.py --tag with_vmask --variant $v 2>&1 | grep -v amdgpu.ids | tee -a $o/vmask_ab.txt
    VTQ_LIB_PATH=tools/_abl/novmask.so python3 tools/attn_probe.py --tag without --variant $v 2>&1 | grep -v amdgpu.ids | tee -a $o/vmask_ab.txt
  done
done
