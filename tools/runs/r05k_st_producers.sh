#!/bin/bash
# small-tile GEMM with DMA-only producer waves beside the MFMA waves (-DVTQ_GEMM_ST_EXPLORE build: tools/_abl/stx.so)
O=gpurun_out/r05k; mkdir -p $O
VTQ_LIB_PATH=$PWD/tools/_abl/stx.so timeout 600 python3 tools/st_bench.py --variants 0 1 20 23 24 2 21 3 22 25 --batches 1 2 3 4 5 > $O/st_producers.txt 2>&1
cat $O/st_producers.txt
