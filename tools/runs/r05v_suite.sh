#!/bin/bash
# GPU suite on the tree with vtq_set_iqa_token (ABI 7) and use_pos_embedding=False; smoke
O=gpurun_out/r05v; mkdir -p $O
timeout 2400 python3 -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/pytest_gpu.txt
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
cat $O/pytest_gpu.txt; tail -2 $O/smoke.txt
