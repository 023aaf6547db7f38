#!/bin/bash
# A/B: does the attention kernel gain from reading Q / K / V out of the cache?  qkvc: the QKV GEMM's epilogue stores with the write-back policy instead of nt;
# qkvcr: additionally the pipelined attention kernel walks its block list from the last sequence to the first (295 MB of planes through a 256 MB Infinity Cache: the rows
# written last are the ones still there); swrev: the reversed walk alone.  Same bits.  tools/class_profile.py, interleaved on one box, B = 32 and B = 16.
cd "$(dirname "$0")/../.."
o=gpurun_out/r05zc; mkdir -p $o
VTQ_LIB_PATH=tools/_abl/qkvcr.so timeout 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "golden and not bench_sizes" 2>&1 | tail -2 | tee $o/pytest.txt
for B in 32 16; do
for r in 1 2 3; do
  for v in shipped qkvc qkvcr swrev; do
    if [ $v = shipped ]; then unset VTQ_LIB_PATH; else export VTQ_LIB_PATH=tools/_abl/$v.so; fi
    echo "## $v B=$B (round $r)" | tee -a $o/classes.txt
    timeout 300 python3 tools/class_profile.py --precision fp16x3 --steps 20 --batch $B 2>&1 | grep -E "ms/step unprofiled|qkv|attention|out_proj" | tee -a $o/classes.txt
  done
done
done
unset VTQ_LIB_PATH
