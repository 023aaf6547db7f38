#!/bin/bash
# Final tree of round 5 (re-run after every later change of the session): GPU suite (product library), the fp8 experiment's tests on its build, smoke, the default bench line
O=gpurun_out/r05z6; mkdir -p $O
timeout 2400 python3 -m pytest tests -m gpu -q 2>&1 | tail -4 > $O/pytest_gpu.txt
VTQ_LIB_PATH=$PWD/vtamiq_amd/libvtamiq_hip_fp8.so timeout 1200 python3 -m pytest tests/test_gpu_fp8.py -m gpu -q 2>&1 | tail -2 > $O/pytest_gpu_fp8_build.txt
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
timeout 900 python3 bench.py > $O/bench_line.json 2> $O/bench_err.txt
cat $O/pytest_gpu.txt $O/pytest_gpu_fp8_build.txt; tail -2 $O/smoke.txt
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r05z6/bench_line.json'))
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["forward_mfma_frac"], "roofline", d["roofline"]["frac"], d["roofline"].get("frac_of_practical"), "e2e", d["e2e"]["frac_of_value"], "secondary", d["secondary"]["value"], d["secondary"]["forward_mfma_frac"])
print([(r["batch"], round(r["ms_per_forward"],3)) for r in d["latency"]["rows"]])
PY
