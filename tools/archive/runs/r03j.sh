#!/bin/bash
# GPU box: kernel tests, then same-box A/B (round-2 tree base.so vs this tree) of the four encoder GEMMs in fp16x3 / fp16 / bf16x3 / fp8
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03j; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_kernels.py -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for rep in 1 2 3 4; do
  VTQ_LIB_PATH=$PWD/tools/_abl/base.so timeout 300 python3 tools/gemm_bench.py --fmt fp16x3 fp16 fp8 --rounds 9 2>&1 | grep flags | sed 's/^/base /' >> $O/gemm.txt
  timeout 300 python3 tools/gemm_bench.py --fmt fp16x3 fp16 fp8 --rounds 9 2>&1 | grep flags | sed 's/^/new  /' >> $O/gemm.txt
done
python3 - <<'PY'
import collections
d=collections.defaultdict(list)
for l in open('gpurun_out/r03j/gemm.txt'):
    t=l.split(); i=t.index('us'); d[(t[0],t[2],t[3])].append((float(t[i-1]), float(l.split('min')[1].split()[0])))
for k in sorted(d, key=lambda k:(k[1],k[2],k[0])):
    v=d[k]; print(k, "median-of-rounds us:", [x[0] for x in v], "mean %.1f"%(sum(x[0] for x in v)/len(v)), " min us:", min(x[1] for x in v))
PY
