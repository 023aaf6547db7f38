#!/bin/bash
# GPU box: L2-miss-side traffic and L2 hit rate of the four encoder GEMMs (fp16x3, fp16): separate --pmc passes, --kernel-trace only beside them
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03m; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
G="python3 $R/tools/gemm_bench.py --rounds 1 --fmt fp16x3 fp16"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o f -- $G > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o w -- $G > $O/write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/tcc -o t -- $G > $O/tcc.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections, sys
sys.path.insert(0,'tools'); import summarize_prof as SP
O='gpurun_out/r03m'
def per_dispatch(d):
    f=glob.glob(f'{O}/{d}/**/*counter_collection.csv', recursive=True)[0]
    rows=collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if 'gemm_pp2' in r['Kernel_Name']:
            rows[int(r['Dispatch_Id'])].setdefault('k', SP.short(r['Kernel_Name'])); rows[int(r['Dispatch_Id'])][r['Counter_Name']]=float(r['Counter_Value'])
            rows[int(r['Dispatch_Id'])]['grid']=r.get('Grid_Size','')
    return [rows[k] for k in sorted(rows)]
fe, wr, tc = per_dispatch('fetch'), per_dispatch('write'), per_dispatch('tcc')
# gemm_bench order per format: qkv, outproj, fc1, fc2, each: 1 warm-up call + 5 timed calls
names=['qkv','outproj','fc1','fc2']
print("# per dispatch (mean over the 6 launches of a shape): FETCH_SIZE x2 (gfx950 correction), WRITE_SIZE, L2 hit rate")
i=0
for fmt in ('fp16x3','fp16'):
    for n in names:
        f=fe[i:i+6]; w=wr[i:i+6]; t=tc[i:i+6]; i+=6
        fm=sum(x['FETCH_SIZE'] for x in f)/len(f)*1024*2/1e6; wm=sum(x['WRITE_SIZE'] for x in w)/len(w)*1024/1e6
        hit=sum(x['TCC_HIT_sum'] for x in t); mis=sum(x['TCC_MISS_sum'] for x in t)
        print(f"{fmt:7s} {n:8s} {f[0]['k']:36s} reads beyond L2 {fm:8.1f} MB   writes {wm:8.1f} MB   L2 hit rate {hit/(hit+mis):.3f}")
PY
