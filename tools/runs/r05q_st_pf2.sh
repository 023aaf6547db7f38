#!/bin/bash
# W-prefetch wave (one tile per XCD and W column tile touches the tile's W lines once): cold and warm, against the shipped shapes
O=gpurun_out/r05q; mkdir -p $O; : > $O/st_pf2.txt
echo "## cold weights (40 buffers)" >> $O/st_pf2.txt
VTQ_LIB_PATH=$PWD/tools/_abl/stx2.so timeout 400 python3 tools/st_bench.py --variants 0 1 26 2 27 3 28 --batches 1 2 4 --cold 40 2>&1 | grep -v amdgpu >> $O/st_pf2.txt
echo "## warm (one weight buffer)" >> $O/st_pf2.txt
VTQ_LIB_PATH=$PWD/tools/_abl/stx2.so timeout 400 python3 tools/st_bench.py --variants 0 1 26 2 27 3 28 --batches 1 2 4 2>&1 | grep -v amdgpu >> $O/st_pf2.txt
cut -c1-420 $O/st_pf2.txt
