#!/bin/bash
# instruction-cache counters of the chain kernel
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_icache_$1; shift
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_IFETCH --output-format csv -d $OUT -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-convergence --batched-tiles 0 "$@" > /dev/null 2> $OUT/err.txt
python3 - <<PY
import csv, glob, collections
for p in glob.glob("$OUT/*/*_counter_collection.csv"):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(p)):
        if 'mpp_chain' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items(): print(k, 'mean=%.4g'%(sum(v)/len(v)))
PY
tail -3 $OUT/err.txt
