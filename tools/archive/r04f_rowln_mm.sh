#!/bin/bash
# MFMA statement forms of the whole-row kernel: pairs interleaved (default) vs one row block per statement; with / without the hazard nops
# (diagnostic builds; the no-nop ones may compute wrong values)
cd "$(dirname "$0")/../.."
o=gpurun_out/r04f; mkdir -p $o
bash tools/build_abl.sh mm1 "-DVTQ_GEMM_DIAG -DVTQ_MEASURE -DVTQ_RL_MM=1" > $o/b1.txt 2>&1 || { tail -5 $o/b1.txt; exit 1; }
bash tools/build_abl.sh mm0 "-DVTQ_GEMM_DIAG -DVTQ_MEASURE -DVTQ_RL_MM=0" > $o/b0.txt 2>&1 || { tail -5 $o/b0.txt; exit 1; }
bash tools/build_abl.sh mm1nn '-DVTQ_GEMM_DIAG -DVTQ_MEASURE -DVTQ_RL_MM=1 -DVTQ_RL_PRE="" -DVTQ_RL_POST=""' > $o/b2.txt 2>&1 || { tail -5 $o/b2.txt; exit 1; }
bash tools/build_abl.sh mm1abl3 "-DVTQ_GEMM_DIAG -DVTQ_MEASURE -DVTQ_RL_MM=1 -DVTQ_RL_ABL=3" > $o/b3.txt 2>&1 || { tail -5 $o/b3.txt; exit 1; }
for v in mm1 mm0 mm1nn mm1abl3; do VTQ_LIB_PATH=tools/_abl/$v.so python3 tools/rowln_probe.py --noln 2>&1 | grep -v amdgpu.ids | tee -a $o/rowln_mm.txt; done
timeout 600 python3 -m pytest tests/test_gpu_kernels.py -q -x -k "rowln" 2>&1 | tail -3
