#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel trace of bench.py: per-forward wall time, per-kernel-type busy time, overlap between the two streams."""
import csv, glob, re, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "").replace("vtq::", "")
    return re.sub(r"\(.*", "", n)[:40]
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Queue_Id"]) for r in rows]
ks.sort()
# split into forwards by pack_patches kernel
starts = [i for i, k in enumerate(ks) if k[2].startswith("pack_patches")]
if len(starts) < 3:
    print("not enough forwards"); sys.exit()
a, b = starts[-2], starts[-1]          # the last complete forward
fw = ks[a:b]
t0, t1 = fw[0][0], max(k[1] for k in fw)
wall = (t1 - t0) / 1e3
busy = collections.defaultdict(float); cnt = collections.Counter()
for s, e, n, q in fw:
    busy[n] += (e - s) / 1e3; cnt[n] += 1
# union of busy intervals (any kernel running)
iv = sorted((s, e) for s, e, _, _ in fw)
union = 0; cs, ce = iv[0]
for s, e in iv[1:]:
    if s > ce: union += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
union += ce - cs
print(f"forward wall {wall:.1f} us; sum of kernel durations {sum(busy.values()):.1f} us; GPU busy (union) {union/1e3:.1f} us; idle {wall-union/1e3:.1f} us; queues {sorted(set(k[3] for k in fw))}")
for n, v in sorted(busy.items(), key=lambda x: -x[1])[:14]:
    print(f"  {n:42s} {cnt[n]:4d} launches  {v:9.1f} us  ({v/sum(busy.values())*100:4.1f}%)  avg {v/cnt[n]:7.1f}")
