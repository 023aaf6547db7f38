#!/bin/bash
# small-tile GEMMs with a W-prefetch wave, cold weights (40 buffers) and warm
O=gpurun_out/r05o; mkdir -p $O
VTQ_LIB_PATH=$PWD/tools/_abl/stx.so timeout 600 python3 tools/st_bench.py --variants 0 1 26 29 30 2 27 3 22 28 --batches 1 2 4 --cold 40 > $O/st_pf_cold.txt 2>&1
VTQ_LIB_PATH=$PWD/tools/_abl/stx.so timeout 600 python3 tools/st_bench.py --variants 0 1 26 29 30 2 27 3 22 28 --batches 1 2 4 > $O/st_pf_warm.txt 2>&1
cut -c1-520 $O/st_pf_cold.txt; echo WARM; cut -c1-520 $O/st_pf_warm.txt
