#!/bin/bash
# GPU box: same-box A/B of the round-2 tree (tools/_abl/base.so) against this tree on the four encoder GEMMs, interleaved, 4 repetitions
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03h; mkdir -p $O
for rep in 1 2 3 4; do
  VTQ_LIB_PATH=$PWD/tools/_abl/base.so timeout 300 python3 tools/gemm_bench.py --fmt fp16x3 fp16 --rounds 9 2>&1 | grep flags | sed 's/^/base /' >> $O/gemm.txt
  timeout 300 python3 tools/gemm_bench.py --fmt fp16x3 fp16 --rounds 9 2>&1 | grep flags | sed 's/^/new  /' >> $O/gemm.txt
done
python3 - <<'PY'
import collections,re
d=collections.defaultdict(list)
for l in open('gpurun_out/r03h/gemm.txt'):
    t=l.split(); d[(t[0],t[2],t[3])].append((float(t[7]), float(t[-5])))
for k in sorted(d, key=lambda k:(k[1],k[2],k[0])):
    v=d[k]; print(k, "median-of-rounds us:", [x[0] for x in v], "mean %.1f"%(sum(x[0] for x in v)/len(v)), " min us:", min(x[1] for x in v))
PY
