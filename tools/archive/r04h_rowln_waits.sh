#!/bin/bash
# where the K loop of the whole-row kernel waits (diagnostic build)
cd "$(dirname "$0")/../.."
o=gpurun_out/r04h; mkdir -p $o
bash tools/build_abl.sh diag "-DVTQ_GEMM_DIAG -DVTQ_MEASURE" > $o/b.txt 2>&1 || { tail -5 $o/b.txt; exit 1; }
VTQ_LIB_PATH=tools/_abl/diag.so python3 tools/rowln_probe.py 2>&1 | grep -v amdgpu.ids | tee -a $o/rowln_waits.txt
