#!/bin/bash
# GPU box: first-round experiments (batch / part-batch sweep, fp16 MFMA subnormal probe, default bench line)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02a; mkdir -p $O
./tools/micro/f16_probe > $O/f16_probe.txt 2>&1
python bench.py > $O/bench_default.json 2> $O/bench_default.err
for parts in 1 2; do for B in 8 16 32 64; do
  VTQ_PARTS=$parts python bench.py --batch $B --steps 10 --warmup 3 --no-cpu-baseline --no-north-star 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('parts=$parts B=$B', 'x3 %.1f pairs/s' % d['value'], 'bf16 %.1f' % d['other_mode']['value'], 'fc1 ms %.4f frac %.4f' % (d['roofline']['avg_launch_ms'], d['roofline']['frac']))
" >> $O/sweep.txt 2>&1
done; done
cat $O/f16_probe.txt $O/sweep.txt; tail -c 1500 $O/bench_default.json
