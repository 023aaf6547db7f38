#!/bin/bash
# the default bench line of this tree (what the driver runs)
O=gpurun_out/r05e; mkdir -p $O
timeout 900 python3 bench.py > $O/bench_line.json 2> $O/bench_err.txt
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r05e/bench_line.json'))
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["forward_mfma_frac"], "of practical", d.get("forward_frac_of_practical"))
rf=d["roofline"]; print({k: rf.get(k) for k in ("achieved","frac","practical_peak_tflops_measured_here","frac_of_practical","mfma_issue_tflops","traffic","traffic_source")})
print(rf.get("practical_peak"))
print("auto", d.get("auto_overhead_measured_in_this_run"))
print("latency"); [print(r) for r in d["latency"]["rows"]]; [print(r) for r in d["latency"]["reference_default_topology"]["rows"]]
print("secondary", d.get("secondary"))
print("e2e", {k: d["e2e"][k] for k in ("value","frac_of_value","loop_frac_of_value")})
print("north_star", {k:(v if not isinstance(v,dict) else (v["value"], v["forward_mfma_frac"])) for k,v in d["north_star_point"].items()})
PY
tail -5 $O/bench_err.txt
