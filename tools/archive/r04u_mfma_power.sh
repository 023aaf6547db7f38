#!/bin/bash
mkdir -p gpurun_out/r04u
timeout 600 tools/micro/mfma_power > gpurun_out/r04u/mfma_power.txt 2>&1
cat gpurun_out/r04u/mfma_power.txt
