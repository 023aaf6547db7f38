#!/bin/bash
# Round 3, GPU box (gpurun -- bash tools/collect_profiles_r03.sh): the rocprofv3 passes and plain runs behind profiles/r03_*.
# Counters in their own passes (--pmc with --kernel-trace only), the program itself after `--`.  Raw output under gpurun_out/r03p/;
# python tools/make_profiles_r03.py (in the repo afterwards) writes the summaries kept under profiles/.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03p; mkdir -p $O
cd $R
python3 bench.py > $O/bench_line.json 2> $O/bench_line.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 $R/bench.py --no-cpu-baseline --no-second-mode --no-north-star --no-fidelity --no-sustained --no-live-traffic > $O/bench_line_headline_profiled.json 2> $O/stats.err
G="python3 $R/tools/gemm_bench.py --only fc1 --rounds 1 --fmt fp16x3 fp16x2 fp16 bf16x3 fp8"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o f -- $G > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o w -- $G > $O/write.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/gemm_pmc -o g -- $G > $O/gemm_pmc.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/gemm_pmc2 -o g -- $G > $O/gemm_pmc2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/attn_pmc -o a -- python3 $R/tools/attn_bench.py --fmt fp16x3 fp16 > $O/attn_pmc.log 2>&1
cd $R
rm -rf $O/stats/*trace* $O/*/*/*kernel_trace* $O/*/*kernel_trace* 2>/dev/null
python3 tools/attn_bench.py --fmt fp16x3 fp16 bf16x3 > $O/attn_bench.txt 2>&1
python3 tools/gemm_bench.py --fmt fp16x3 fp16x2 fp16 bf16x3 bf16 fp8 > $O/gemm_bench.txt 2>&1
VTQ_GEMM_FLAGS=8 python3 tools/gemm_bench.py --fmt fp16x3 fp16 > $O/gemm_bench_noepi.txt 2>&1
python3 tools/class_profile.py > $O/class_profile.txt 2>&1
python3 tools/run_config.py --variant ViT-L16 --batch 16 --patches 1024 --scales 3 > $O/config3_vitl.txt 2>&1
python3 tools/run_config.py --variant ViT-B16 --batch 16 --patches 512 --scales 5 --refdefault > $O/refdefault.txt 2>&1
python3 tools/run_config.py --variant ViT-B16 --batch 4 --patches 2500 --scales 1 > $O/n2500.txt 2>&1
python3 -m pytest tests/test_gpu_fp8.py -q -s > $O/fp8_tests.txt 2>&1
python3 tools/fuzz_parity.py --cases 150 --precision fp16x3 > $O/fuzz.txt 2>&1
ls $O
