#!/bin/bash
# GPU box: what bounds the 16-bit GEMM epilogue -- diagnostic builds with parts of it removed (results then wrong by design):
#   a1 no GELU arithmetic, a2 no copy-out (LDS read + global store), a3 no global store (LDS read kept), a4 no LDS staging and no copy-out
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03i; mkdir -p $O
for v in diag diag_a1 diag_a2 diag_a3 diag_a4 diag; do
  echo "== build $v" >> $O/abl.txt
  VTQ_LIB_PATH=$PWD/tools/_abl/$v.so timeout 300 python3 tools/clock_probe.py --fmt fp16x3 fp16 --only fc1 qkv --shadow 0 --warm 1.0 2>&1 | grep -v amdgpu >> $O/abl.txt
done
cat $O/abl.txt
