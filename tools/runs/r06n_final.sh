#!/bin/bash
# final tree of round 6 (attention rule: no split form, fill threshold 0.74): full GPU suite, smoke, bench line
cd "$(dirname "$0")/../.."
O=gpurun_out/r06n; mkdir -p $O
timeout 2400 python3 -m pytest tests -m gpu -q 2>&1 | tail -4 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 900 python3 bench.py > $O/bench_line.json 2> $O/bench_stderr.txt
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r06n/bench_line.json"))
print("value", d["value"], "ms/step", d["ms_per_step"], "roofline", d["roofline"]["frac"], d["roofline"].get("avg_launch_ms"))
print("e2e", d["e2e"]["frac_of_value"], "secondary", d["secondary"]["value"])
print("latency", [(r["batch"], round(r["ms_per_forward"], 3)) for r in d["latency"]["rows"]])
print("long", [(r["patches"], round(r["ms_per_forward"], 2), round(r["forward_mfma_frac"], 4)) for r in d["long_sequence"]["rows"]])
PY
