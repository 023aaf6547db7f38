#!/bin/bash
# wider fuzz of the parity mode on the final tree: second and third seeds
mkdir -p gpurun_out/r04y
timeout 1500 python3 tools/fuzz_parity.py --cases 300 --seed 1 --precision fp16x3 > gpurun_out/r04y/fuzz_seed1.txt 2>&1
timeout 900 python3 tools/fuzz_parity.py --cases 100 --seed 2 > gpurun_out/r04y/fuzz_seed2_drawn_modes.txt 2>&1
tail -3 gpurun_out/r04y/fuzz_seed1.txt gpurun_out/r04y/fuzz_seed2_drawn_modes.txt
