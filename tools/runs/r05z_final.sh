#!/bin/bash
# final collection of round 5 on the final tree
O=gpurun_out/r05z; mkdir -p $O
timeout 2400 python3 -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/pytest_gpu.txt
VTQ_LIB_PATH=$PWD/vtamiq_amd/libvtamiq_hip_fp8.so timeout 1200 python3 -m pytest tests/test_gpu_fp8.py -m gpu -q 2>&1 | tail -3 > $O/pytest_gpu_fp8_build.txt
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
VTQ_LIB_PATH=$PWD/tools/_abl/stx.so timeout 900 python3 tools/st_bench.py --variants 0 1 9 2 10 3 18 4 5 6 --batches 1 2 3 4 5 6 8 12 16 32 --json $O/st_warm.json > $O/st_warm.txt 2>&1
VTQ_LIB_PATH=$PWD/tools/_abl/stx.so timeout 900 python3 tools/st_bench.py --variants 0 1 9 2 10 3 18 --batches 1 2 3 4 5 6 8 --cold 40 > $O/st_cold.txt 2>&1
VTQ_LIB_PATH=$PWD/tools/_abl/stx.so timeout 300 python3 tools/st_bench.py --variants 0 9 11 12 13 14 15 18 16 17 --batches 1 4 --only outproj fc2 > $O/st_ablation.txt 2>&1
for f in fp16 fp16x2 bf16x3 bf16; do timeout 300 python3 tools/st_bench.py --variants 0 1 2 3 --batches 1 2 4 --fmt $f > $O/st_$f.txt 2>&1; done
timeout 600 python3 tools/small_batch.py --classes --json $O/after.json > $O/after.txt 2>&1
timeout 300 python3 tools/small_batch.py --refdefault --batches 1 16 --patches 512 --classes --json $O/after_refdefault.json > $O/after_refdefault.txt 2>&1
timeout 400 python3 tools/gemm_stress.py --seconds 200 --seed 31 > $O/gemm_stress.txt 2>&1
timeout 300 python3 tools/attn_stress.py --seconds 120 --seed 32 > $O/attn_stress.txt 2>&1
timeout 400 python3 tools/forward_stress.py --seconds 200 --seed 33 > $O/forward_stress.txt 2>&1
timeout 900 python3 bench.py > $O/bench_line.json 2> $O/bench_err.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_bench -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-second-mode --no-north-star --no-live-traffic --no-sustained --no-fidelity --no-e2e --no-secondary --no-latency --no-practical-peak --no-auto-overhead --no-collective-check > $GRAFT_REPO_ROOT/$O/bench_line_headline_profiled.json 2>/dev/null
cd $GRAFT_REPO_ROOT
python3 tools/summarize_prof.py stats $O/prof_bench > $O/bench_headline_kernel_stats.txt 2>&1; rm -rf $O/prof_bench
cat $O/pytest_gpu.txt $O/pytest_gpu_fp8_build.txt; tail -1 $O/smoke.txt; tail -1 $O/gemm_stress.txt; tail -1 $O/attn_stress.txt; tail -1 $O/forward_stress.txt; grep -c DIFFERENT $O/st_*.txt
grep -E "^ +[0-9]+ " $O/after.txt $O/after_refdefault.txt; head -c 600 $O/bench_line.json; echo; head -8 $O/bench_headline_kernel_stats.txt
