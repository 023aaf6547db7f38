#!/bin/bash
# GPU box: full -m gpu suite and the bench line on the current tree
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03l; mkdir -p $O
timeout 1800 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
timeout 900 python3 bench.py > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err; python3 -c "
import json;d=json.loads([l for l in open('$O/bench.json') if l.startswith('{')][-1]);print({k:d[k] for k in ('value','ms_per_step')}, d['sustained']['value'], d['roofline']['frac'], {k:v['value'] for k,v in d['other_modes'].items()}, {k:round(v['forward_mfma_frac'],4) for k,v in d['north_star_point'].items() if isinstance(v,dict)}); print({k:(round(v['SROCC'],5), round(v['PLCC'],5)) for k,v in d['fidelity']['modes'].items()})"
