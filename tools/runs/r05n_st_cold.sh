#!/bin/bash
# small-tile GEMMs with COLD weights (40 weight buffers cycled: every launch's W comes from HBM, as inside a forward) and the forward's small-batch table
O=gpurun_out/r05n; mkdir -p $O
VTQ_LIB_PATH=$PWD/tools/_abl/stx.so timeout 600 python3 tools/st_bench.py --variants 0 1 9 2 10 3 18 --batches 1 2 4 --cold 40 > $O/st_cold.txt 2>&1
timeout 600 python3 tools/small_batch.py --classes --batches 1 2 4 8 --json $O/after.json > $O/after.txt 2>&1
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "tile_shapes or golden or ragged" 2>&1 | tail -3 > $O/pytest_parity.txt
cat $O/st_cold.txt | cut -c1-400; grep -E "^ +[0-9]+ |fc2|out_proj|qkv|fc1" $O/after.txt; cat $O/pytest_parity.txt
