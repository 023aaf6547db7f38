#!/bin/bash
# final tree of round 6: the new kernel tests, randomised bit-for-bit stress (attention: 4-wave / pipelined / split, both block walks; whole forward; parity fuzz),
# and rocprofv3 --kernel-trace --stats of the headline-only bench command with the line the same command printed
cd "$(dirname "$0")/../.."
O=gpurun_out/r06i; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "attention_kernels_agree or cu_partition" 2>&1 | tail -3 | tee $O/pytest_new.txt
timeout 300 python3 tools/attn_stress.py --seconds 200 --seed 61 > $O/attn_stress.txt 2>&1; tail -2 $O/attn_stress.txt
timeout 400 python3 tools/forward_stress.py --seconds 240 --seed 62 > $O/forward_stress.txt 2>&1; tail -1 $O/forward_stress.txt
timeout 600 python3 tools/fuzz_parity.py --seed 63 > $O/fuzz.txt 2>&1; tail -3 $O/fuzz.txt
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_bench -o p -- python3 $R/bench.py --no-cpu-baseline --no-second-mode --no-north-star --no-live-traffic --no-sustained --no-fidelity --no-e2e --no-secondary --no-latency --no-long-sequence --no-practical-peak --no-auto-overhead --no-collective-check > $R/$O/bench_line_headline_profiled.json 2>/dev/null
cd $R
python3 tools/summarize_prof.py stats $O/prof_bench > $O/bench_headline_kernel_stats.txt 2>&1; rm -rf $O/prof_bench
head -14 $O/bench_headline_kernel_stats.txt; python3 -c "
import json; d=json.load(open('$O/bench_line_headline_profiled.json')); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
