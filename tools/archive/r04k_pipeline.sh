#!/bin/bash
# the loader-side pipeline as product code: test + the bench's e2e block through it
cd "$(dirname "$0")/../.."
o=gpurun_out/r04k; mkdir -p $o
timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -x -k "pipeline or fused or patch_sample" 2>&1 | tail -3
python3 bench.py --no-cpu-baseline --no-fidelity --no-north-star --no-second-mode --no-sustained --no-live-traffic --no-secondary > $o/bench_e2e.txt 2> $o/err.txt
python3 -c "
import json; d=json.loads(open('$o/bench_e2e.txt').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step']); e=d['e2e']; print({k:e[k] for k in ('value','value_loop_only','ms_per_batch','reductions_seconds','frac_of_value','loop_frac_of_value')})"
