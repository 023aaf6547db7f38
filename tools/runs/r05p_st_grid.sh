#!/bin/bash
# XCD grid of the small-tile kernels (rows x columns of the 8 XCDs over the tile grid), cold weights: forced grids against the host's choice
O=gpurun_out/r05p; mkdir -p $O; : > $O/st_grid.txt
for g in 129 66 36 24 0; do
  echo "## VTQ_ST_GRID=$g (rows*16+cols; 0 = the host's choice)" >> $O/st_grid.txt
  VTQ_ST_GRID=$g VTQ_LIB_PATH=$PWD/tools/_abl/stx.so timeout 300 python3 tools/st_bench.py --variants 1 2 3 --batches 1 2 3 4 --cold 40 2>&1 | grep -v amdgpu >> $O/st_grid.txt
done
cut -c1-330 $O/st_grid.txt
timeout 600 python3 tools/small_batch.py --classes --batches 1 2 4 --json $O/after.json > $O/after.txt 2>&1
grep -E "^ +[0-9]+ |fc2|out_proj|qkv|fc1|patch" $O/after.txt
