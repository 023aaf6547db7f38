#!/bin/bash
# the final small-tile kernels (with producer waves): tests that touch the GEMMs, the tile-shape table, the small-batch table
O=gpurun_out/r05l; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "gemm" 2>&1 | tail -3 > $O/pytest_gemm.txt
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "tile_shapes or full_size or golden or repeated or ragged" 2>&1 | tail -3 > $O/pytest_parity.txt
VTQ_LIB_PATH=$PWD/tools/_abl/stx.so timeout 900 python3 tools/st_bench.py --variants 0 1 9 2 10 3 18 4 5 6 --batches 1 2 3 4 5 6 8 12 16 32 --json $O/st_final.json > $O/st_final.txt 2>&1
VTQ_LIB_PATH=$PWD/tools/_abl/stx.so timeout 300 python3 tools/st_bench.py --variants 0 9 11 12 13 14 15 18 16 17 --batches 1 4 --only outproj fc2 > $O/st_ablation.txt 2>&1
for f in fp16 fp16x2 bf16x3 bf16; do timeout 300 python3 tools/st_bench.py --variants 0 1 2 3 --batches 1 2 4 --fmt $f > $O/st_$f.txt 2>&1; done
timeout 300 python3 tools/gemm_stress.py --seconds 150 --seed 21 > $O/gemm_stress.txt 2>&1
timeout 600 python3 tools/small_batch.py --classes --json $O/after.json > $O/after.txt 2>&1
timeout 300 python3 tools/small_batch.py --refdefault --batches 1 16 --patches 512 --classes --json $O/after_refdefault.json > $O/after_refdefault.txt 2>&1
cat $O/pytest_gemm.txt $O/pytest_parity.txt; tail -1 $O/gemm_stress.txt; grep -c DIFFERENT $O/st_*.txt; grep -E "^ +[0-9]+ " $O/after.txt $O/after_refdefault.txt
