#!/bin/bash
# phases of the whole-row kernel (diagnostic build) + A/B on the shipped build
cd "$(dirname "$0")/../.."
o=gpurun_out/r04d; mkdir -p $o
bash tools/build_abl.sh diag "-DVTQ_GEMM_DIAG -DVTQ_MEASURE" > $o/build.txt 2>&1 || { tail -5 $o/build.txt; exit 1; }
for extra in "" "--noln"; do VTQ_LIB_PATH=tools/_abl/diag.so python3 tools/rowln_probe.py $extra 2>&1 | grep -v amdgpu.ids | tee -a $o/rowln_probe.txt; done
VTQ_LIB_PATH=tools/_abl/diag.so python3 tools/rowln_probe.py --M 64256 2>&1 | grep -v amdgpu.ids | tee -a $o/rowln_probe.txt
python3 tools/rowln_bench.py --M 32256 --rounds 5 2>&1 | grep -v amdgpu.ids | tee $o/rowln_bench.txt
