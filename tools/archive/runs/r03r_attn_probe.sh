#!/bin/bash
# attention, round 3: sustained time of the two kernels, in-kernel clock + phase spans (diagnostic build), skeleton ablations of the
# pipelined kernel (tools/build_abl.sh adiag "-DVTQ_ATTN_DIAG"; pnf/pnm/pnn/k1..k7 = the -DVTQ_SW_* switches; see profiles/r03_attention_anatomy.txt)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
o=gpurun_out/r03r_attn_probe.txt; : > $o
timeout 300 python3 tools/attn_ab.py >> $o 2>&1
for v in 0 1; do
python3 tools/attn_probe.py --tag shipped --variant $v >> $o 2>&1
VTQ_LIB_PATH=tools/_abl/adiag.so python3 tools/attn_probe.py --tag diag --variant $v >> $o 2>&1
done
for n in pnf pnm pnn k1 k2 k3 k4 k5 k6 k7; do
test -f tools/_abl/$n.so && VTQ_LIB_PATH=tools/_abl/$n.so python3 tools/attn_probe.py --tag $n --variant 1 --fmt fp16x3 >> $o 2>&1
done
grep -v amdgpu.ids $o
