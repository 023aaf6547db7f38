#!/bin/bash
# full GPU suite (fp8 tests in the default invocation), smoke, and the bench line on the tree with the round-6 attention kernel, the long_sequence
# block and the two-pass e2e with deferred fits
cd "$(dirname "$0")/../.."
O=gpurun_out/r06h; mkdir -p $O
timeout 2400 python3 -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/pytest_gpu.txt
cat $O/pytest_gpu.txt
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 900 python3 bench.py > $O/bench_line.json 2> $O/bench_stderr.txt; tail -3 $O/bench_stderr.txt
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r06h/bench_line.json"))
print("value", d["value"], "ms/step", d["ms_per_step"], "roofline", d["roofline"]["frac"], d["roofline"].get("avg_launch_ms"))
print("e2e", {k: d["e2e"].get(k) for k in ("value", "frac_of_value", "loop_frac_of_value", "reductions_seconds", "sets")})
print("long", d.get("long_sequence"))
print("latency", [(r["batch"], round(r["ms_per_forward"], 3)) for r in d["latency"]["rows"]])
print("north", d.get("north_star_point"))
PY
