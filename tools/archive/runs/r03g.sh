#!/bin/bash
# GPU box: fp8 tests after the check fix; kernel tests; A/B of the round-2 tree (base.so) against this tree (bias in the accumulators through an LDS bias image,
# hand-ordered 4-wide GELU, 6-instruction hi/lo split) on the four encoder GEMMs; bench line
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03g; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_fp8.py -x -q > $O/pytest.txt 2>&1; tail -6 $O/pytest.txt
for rep in 1 2 3; do
  VTQ_LIB_PATH=$PWD/tools/_abl/base.so timeout 300 python3 tools/gemm_bench.py --fmt fp16x3 fp16 --rounds 9 2>&1 | grep flags | sed 's/^/base /' >> $O/gemm.txt
  timeout 300 python3 tools/gemm_bench.py --fmt fp16x3 fp16 --rounds 9 2>&1 | grep flags | sed 's/^/new  /' >> $O/gemm.txt
done
sort -k1,1 -k3,4 -s $O/gemm.txt
timeout 300 python3 tools/gemm_bench.py --fmt fp8 bf16x3 --rounds 5 2>&1 | grep flags
timeout 900 python3 bench.py --no-fidelity --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python3 -c "
import json;d=json.loads([l for l in open('$O/bench.json') if l.startswith('{')][-1]);print({k:d[k] for k in ('value','ms_per_step')}, d['sustained']['value'], d['roofline']['frac'], {k:v['value'] for k,v in d['other_modes'].items()}, {k:v['forward_mfma_frac'] for k,v in d['north_star_point'].items() if isinstance(v,dict)})"
