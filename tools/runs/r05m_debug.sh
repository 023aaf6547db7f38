#!/bin/bash
O=gpurun_out/r05m; mkdir -p $O
timeout 1500 python3 -X faulthandler -m pytest tests/test_gpu_parity.py -m gpu -q -x -v -k "tile_shapes or full_size or golden or repeated or ragged" > $O/pytest_parity.txt 2>&1
tail -40 $O/pytest_parity.txt
