#!/bin/bash
# A/B: issue priority alternating between the two waves of a SIMD in the pipelined attention kernel (-DVTQ_SW_PRIO=1 per phase, =2 per fragment group,
# =3 static for waves 4-7) against shipped.  Same arithmetic (s_setprio only).  Stamps (diagnostic builds) and sustained time, interleaved on one box.
cd "$(dirname "$0")/../.."
o=gpurun_out/r05z8; mkdir -p $o
for v in adiag swprio1d swprio2d swprio3d; do
  VTQ_LIB_PATH=tools/_abl/$v.so timeout 200 python3 tools/attn_probe.py --variant 1 --fmt fp16x3 2>&1 | grep -v amdgpu | tee -a $o/stamps.txt
done
for r in 1 2 3; do
  for v in shipped swprio1 swprio2 swprio3; do
    if [ $v = shipped ]; then unset VTQ_LIB_PATH; else export VTQ_LIB_PATH=tools/_abl/$v.so; fi
    timeout 200 python3 tools/attn_probe.py --variant 1 --fmt fp16x3 --tag $v 2>&1 | grep -v amdgpu | tee -a $o/sustained.txt
  done
done
unset VTQ_LIB_PATH
