#!/bin/bash
# instruction-cache and issue counters of the deep-round kernel on the bench tile (one 512x512 tile, 100 001 steps)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_deep_icache_$1; shift
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_IFETCH --output-format csv -d $OUT -- python3 profiles/tools/deep_probe.py --reps 1 --configs "$@" > $OUT/out.txt 2> $OUT/err.txt
python3 - <<PY
import csv, glob, collections
for p in glob.glob("$OUT/*/*_counter_collection.csv"):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(p)):
        if 'mpp_deep' in r['Kernel_Name'] or 'mpp_chain' in r['Kernel_Name']: agg[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
    for kn, d in agg.items():
        print(kn)
        for k,v in d.items(): print('   ', k, 'mean=%.5g'%(sum(v)/len(v)), 'n=%d'%len(v))
PY
tail -3 $OUT/err.txt
