#!/bin/bash
# Two model instances on two streams at the HEADLINE batch sizes: does a second in-flight batch fill the tails of the first's launches (GEMM remainder
# rounds, attention seams, the head's dependent chain)?  Against one instance and against the same pairs batched into one forward.
O=gpurun_out/r05u; mkdir -p $O
timeout 600 python3 tools/concurrent_streams.py --batch 16 --streams 1 2 --steps 30 > $O/streams.txt 2>&1
timeout 600 python3 tools/concurrent_streams.py --batch 32 --streams 1 2 --steps 30 >> $O/streams.txt 2>&1
timeout 600 python3 tools/concurrent_streams.py --batch 32 --streams 1 2 --steps 30 >> $O/streams.txt 2>&1
grep -v amdgpu $O/streams.txt
