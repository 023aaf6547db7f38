#!/bin/bash
# kernel timeline of one B = 1 forward (durations and gaps between dispatches)
O=gpurun_out/r05i; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/tr_b1 -o p -- python3 $GRAFT_REPO_ROOT/tools/small_batch.py --batches 1 --steps 20 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/tr_ref -o p -- python3 $GRAFT_REPO_ROOT/tools/small_batch.py --batches 1 --steps 20 --refdefault --patches 512 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/trace_forward.py $O/tr_b1 > $O/trace_b1.txt 2>&1
python3 tools/trace_forward.py $O/tr_ref > $O/trace_refdefault_b1.txt 2>&1
rm -rf $O/tr_b1 $O/tr_ref
cat $O/trace_b1.txt | tail -150
