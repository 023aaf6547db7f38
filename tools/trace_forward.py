#!/usr/bin/env python3
"""One forward's kernel timeline from a rocprofv3 --kernel-trace CSV: every dispatch in start order with its duration and the gap to the end of
the previous one -- where a small-batch forward's time goes between kernels (launch latency chain) as opposed to inside them.
    rocprofv3 --kernel-trace --output-format csv -d DIR -o p -- python3 tools/small_batch.py --batches 1 --steps 20
    python3 tools/trace_forward.py DIR [--forward K]"""
import argparse, csv, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import summarize_prof as SP

ap = argparse.ArgumentParser()
ap.add_argument("dir")
ap.add_argument("--forward", type=int, default=-3, help="which forward of the trace (index into the forwards found; default: third from the end)")
ap.add_argument("--first", default="pack_patches_kernel", help="kernel that opens a forward")
a = ap.parse_args()
f = glob.glob(a.dir + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if SP.short(r["Kernel_Name"]).startswith(a.first)]
i0 = starts[a.forward]
i1 = starts[a.forward + 1] if a.forward + 1 < 0 and a.forward + 1 + len(starts) < len(starts) else (starts[starts.index(i0) + 1] if starts.index(i0) + 1 < len(starts) else len(rows))
sel = rows[i0:i1]
t0 = int(sel[0]["Start_Timestamp"])
prev_end = t0
tot_k = tot_gap = 0
agg = {}
print(f"# forward = dispatches {i0}..{i1 - 1} of {os.path.basename(f)}")
print(f"{'#':>3s} {'start_us':>9s} {'dur_us':>8s} {'gap_us':>7s}  kernel (grid x block)")
for n, r in enumerate(sel):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = SP.short(r["Kernel_Name"])
    gap = (s - prev_end) / 1e3
    dur = (e - s) / 1e3
    tot_k += dur; tot_gap += max(gap, 0.0)
    k = agg.setdefault(name, [0, 0.0, 0.0]); k[0] += 1; k[1] += dur; k[2] += max(gap, 0.0)
    wg = r.get("Workgroup_Size_X") or r.get("Workgroup_Size") or "?"
    gx = r.get("Grid_Size_X") or r.get("Grid_Size") or "?"
    print(f"{n:3d} {(s - t0) / 1e3:9.1f} {dur:8.1f} {gap:7.1f}  {name} ({gx} x {wg})")
    prev_end = max(prev_end, e)
span = (prev_end - t0) / 1e3
print(f"# span {span:.1f} us = kernels {tot_k:.1f} us + gaps {tot_gap:.1f} us ({len(sel)} dispatches, mean gap {tot_gap / max(1, len(sel) - 1):.2f} us)")
print("# by kernel: calls, total duration, total gap in front")
for name, (c, d, g) in sorted(agg.items(), key=lambda kv: -kv[1][1] - kv[1][2]):
    print(f"#   {name:60s} {c:3d} {d:8.1f} us {g:8.1f} us")
