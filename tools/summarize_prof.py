#!/usr/bin/env python3
"""Turn rocprofv3 CSV output (kernel_stats / counter_collection) into the small text summaries kept under profiles/."""
import collections, csv, glob, re, sys

def demangle(n):
    """_ZN3vtq12_GLOBAL__N_1<len><name>I<template args>E... -> name<args> for the argument kinds these kernels use
    (llvm-cxxfilt in this ROCm does not know DF16_ = _Float16, so rocprofv3 prints such names mangled)."""
    m = re.match(r"_ZN3vtq12_GLOBAL__N_1(\d+)", n)
    if not m:
        return n
    ln = int(m.group(1)); i = m.end(); name = n[i:i + ln]; i += ln
    args = []
    if i < len(n) and n[i] == "I":
        i += 1
        while i < len(n) and n[i] != "E":
            if n.startswith("DF16_", i): args.append("f16"); i += 5
            elif n.startswith("DF16b", i): args.append("bf16"); i += 5
            elif n.startswith("NS_2f8E", i): args.append("f8"); i += 7
            elif n[i] == "L":
                j = n.index("E", i); v = n[i + 2:j]; args.append(v.replace("n", "-")); i = j + 1
            else: args.append("?"); break
    return f"{name}<{', '.join(args)}>" if args else name


def short(n):
    n = demangle(n)
    n = re.sub(r"void vtq::\(anonymous namespace\)::", "", n)
    n = re.sub(r"vtq::\(anonymous namespace\)::", "", n)
    n = n.replace("<bool _Accum, int, ELi, E>", "<bf16, .., ..> (name garbled by rocprofv3)").replace("<bool _Accum, int, E>", "<bf16, ..> (name garbled by rocprofv3)")
    return re.sub(r"\(.*", "", n)[:64]

def stats(d):
    f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"# rocprofv3 --kernel-trace --stats  ({f.split('/')[-1]}); total kernel time {tot/1e6:.3f} ms")
    print(f"{'kernel':66s} {'calls':>6s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'total_ms':>9s} {'pct':>6s}")
    for r in rows:
        if float(r["TotalDurationNs"]) / tot < 0.0005:
            continue
        print(f"{short(r['Name']):66s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:9.1f} {float(r['MinNs'])/1e3:9.1f} "
              f"{float(r['MaxNs'])/1e3:9.1f} {float(r['TotalDurationNs'])/1e6:9.3f} {float(r['TotalDurationNs'])/tot*100:6.2f}")

def pmc(d, pat):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
    for r in csv.DictReader(open(f)):
        if pat not in r["Kernel_Name"]:
            continue
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
    print(f"# rocprofv3 --pmc ({f.split('/')[-1]}), mean per dispatch")
    for k, v in agg.items():
        for c, val in v.items():
            print(f"{k:60s} {c:28s} {val/cnt[k][c]:18.1f}  (dispatches {cnt[k][c]})")

if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    else:
        pmc(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "")
