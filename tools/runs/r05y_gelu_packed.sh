#!/bin/bash
# A/B: the GELU polynomial of the fc1 epilogue as packed fp32 (v_pk_fma_f32; -DVTQ_GELU_PACKED=1: 30 vector instructions per 4 values instead
# of 44, the same operations on the same values = the same bits) against the shipped scalar form.  Interleaved on one box: the fc1 GEMM alone
# (tools/gemm_bench.py, with its fp64 check) and the whole step (bench.py), plus a bitwise comparison of the two libraries' scores.
cd "$(dirname "$0")/../.."
o=gpurun_out/r05y; mkdir -p $o
tools/micro/valu_issue > $o/valu_issue.txt 2>&1
for r in 1 2 3; do
  for v in shipped gelupk; do
    if [ $v = shipped ]; then unset VTQ_LIB_PATH; else export VTQ_LIB_PATH=tools/_abl/$v.so; fi
    echo "## $v (round $r)" | tee -a $o/gemm.txt
    timeout 300 python3 tools/gemm_bench.py --fmt fp16x3 --only fc1 --rounds 7 2>&1 | grep -v amdgpu.ids | tee -a $o/gemm.txt
  done
done
for r in 1 2 3; do
  for v in shipped gelupk; do
    if [ $v = shipped ]; then unset VTQ_LIB_PATH; else export VTQ_LIB_PATH=tools/_abl/$v.so; fi
    echo "## $v (round $r)" | tee -a $o/bench.txt
    timeout 300 python3 bench.py --no-cpu-baseline --no-fidelity --no-secondary --no-e2e --no-north-star --no-live-traffic --no-collective-check --no-second-mode --no-latency --no-practical-peak --no-auto-overhead 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])" | tee -a $o/bench.txt
  done
done
unset VTQ_LIB_PATH
# same bits: the golden-sized forward through both libraries
for v in shipped gelupk; do
  if [ $v = shipped ]; then unset VTQ_LIB_PATH; else export VTQ_LIB_PATH=tools/_abl/$v.so; fi
  python3 - > $o/scores_$v.txt <<'PY'
import torch, sys, os
sys.path.insert(0, os.getcwd())
from vtamiq_amd import VTAMIQ, synth
m = VTAMIQ(vit_config=dict(variant="ViT-B16", pretrained=False), precision="fp16x3")
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(m.spec, 0).items()}); m = m.cuda().eval()
p, q, _ = synth.make_inputs(m.spec, 8, 500, 3)
tp, tq = torch.from_numpy(p).cuda(), torch.from_numpy(q).cuda()
with torch.no_grad():
    s = m((tp[:, 0].contiguous(), tp[:, 1].contiguous()), (tq[:, 0].contiguous(), tq[:, 1].contiguous()), (None, None))[0]
print(" ".join(f"{v:.9e}" for v in s.cpu().tolist()))
PY
done
cmp $o/scores_shipped.txt $o/scores_gelupk.txt && echo "scores: bit-identical" | tee -a $o/bench.txt
