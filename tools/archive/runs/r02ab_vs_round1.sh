#!/bin/bash
# same-box A/B: round-1 tree (_r01/, built from commit ff1935c) against the current tree: GEMM micro-bench and the bench line
# setup (in the repo, before the gpurun call): mkdir -p _r01 && git archive ff1935c | tar -x -C _r01 && (cd _r01 && python -m vtamiq_amd.build)   -- _r01/ is git-ignored
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02ab; mkdir -p $O
for rep in 1 2; do
  cd $R/_r01 && python3 tools/gemm_bench.py --M 32256 --only qkv outproj fc1 fc2 2>&1 | grep -v amdgpu > $O/r01_gemm_$rep.txt
  cd $R && python3 tools/gemm_bench.py --fmt bf16 bf16x3 fp16x3 2>&1 | grep flags > $O/r02_gemm_$rep.txt
done
cd $R/_r01 && python3 bench.py --no-cpu-baseline > $O/r01_bench.json 2> $O/r01_bench.err
cd $R && python3 bench.py --no-cpu-baseline --no-north-star > $O/r02_bench.json 2> $O/r02_bench.err
cd $R && python3 bench.py --no-cpu-baseline --no-north-star --precision bf16x3 --no-second-mode > $O/r02_bench_bf16x3.json 2> $O/r02_bench_bf16x3.err
cd $R/_r01 && python3 bench.py --no-cpu-baseline > $O/r01_bench2.json 2> $O/r01_bench2.err
cat $O/r01_gemm_1.txt $O/r02_gemm_1.txt $O/r01_gemm_2.txt $O/r02_gemm_2.txt
for f in r01_bench r02_bench r02_bench_bf16x3 r01_bench2; do python3 -c "
import json,sys
d=json.loads([l for l in open('$O/$f.json') if l.startswith('{')][-1]); print('$f', d['config']['numerics'], round(d['value'],1), round(d['ms_per_step'],3), d.get('roofline',{}).get('avg_launch_ms'))"; done
