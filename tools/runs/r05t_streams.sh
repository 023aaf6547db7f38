#!/bin/bash
O=gpurun_out/r05t; mkdir -p $O
timeout 600 python3 tools/concurrent_streams.py --batch 1 > $O/streams.txt 2>&1
timeout 600 python3 tools/concurrent_streams.py --batch 2 --streams 1 2 4 >> $O/streams.txt 2>&1
grep -v amdgpu $O/streams.txt
