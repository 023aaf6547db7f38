#!/bin/bash
# whole-row GEMM with (SGPR base + VGPR offset) LDS-DMA requests: tests, stand-alone A/B, phases
cd "$(dirname "$0")/../.."
o=gpurun_out/r04o; mkdir -p $o
timeout 900 python3 -m pytest tests/test_gpu_kernels.py -q -x -k "rowln" 2>&1 | tail -2
python3 tools/rowln_bench.py --M 32256 --rounds 7 2>&1 | grep -v amdgpu.ids | tee $o/rowln_bench.txt
bash tools/build_abl.sh diag "-DVTQ_GEMM_DIAG -DVTQ_MEASURE" > $o/b.txt 2>&1 || { tail -5 $o/b.txt; exit 1; }
VTQ_LIB_PATH=tools/_abl/diag.so python3 tools/rowln_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-420 | tee $o/rowln_probe.txt
for r in 1 2; do
python3 tools/class_profile.py --precision fp16x3 2>&1 | grep -v amdgpu.ids | grep "ms/step unprofiled\|out_proj\|fc2\|layernorm" | tee -a $o/class_ab.txt
python3 tools/class_profile.py --precision fp16x3 --fused-ln 2>&1 | grep -v amdgpu.ids | grep "ms/step unprofiled\|out_proj\|fc2\|layernorm" | tee -a $o/class_ab.txt
done
