#!/bin/bash
# GPU box: where the GEMM epilogue's cycles go (diagnostic build: s_memtime stamps inside the epilogue) and the order of the two
# steps of an epilogue interval in the two wave groups (shipped: opposite; o1: both copy first; o2: both convert first).
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03b; mkdir -p $O
for v in diag diag_o1 diag_o2 diag; do
  echo "== build $v" >> $O/epi.txt
  VTQ_LIB_PATH=$PWD/tools/_abl/$v.so timeout 300 python3 tools/clock_probe.py --fmt fp16x3 fp16 --only fc1 qkv --shadow 0 --warm 1.5 >> $O/epi.txt 2>&1
done
cat $O/epi.txt
