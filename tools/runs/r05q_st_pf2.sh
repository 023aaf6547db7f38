#!/bin/bash
O=gpurun_out/r05q; mkdir -p $O
for g in 0 24 36; do echo "## VTQ_ST_GRID=$g" >> $O/st_pf2.txt; VTQ_ST_GRID=$g VTQ_LIB_PATH=$PWD/tools/_abl/stx.so timeout 300 python3 tools/st_bench.py --variants 1 26 29 3 28 30 --batches 1 2 4 --cold 40 2>&1 | grep -v amdgpu >> $O/st_pf2.txt; done
cut -c1-400 $O/st_pf2.txt
