#!/bin/bash
# occupancy / issue counters of a many-tile launch
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_batched_$1; shift
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/err.txt
python3 - <<PY
import csv, glob, collections, json
dur = {}
for p in glob.glob("$OUT/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(p)):
        if 'mpp_chain' in r['Kernel_Name']:
            dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp']), r['Grid_Size'] if 'Grid_Size' in r else '')
for p in glob.glob("$OUT/*/*_counter_collection.csv"):
    agg = collections.defaultdict(dict)
    for r in csv.DictReader(open(p)):
        if 'mpp_chain' in r['Kernel_Name']: agg[r['Dispatch_Id']][r['Counter_Name']] = float(r['Counter_Value'])
    for d, v in agg.items():
        ns = dur.get(d, (0, ''))[0]
        print('dispatch', d, 'ns', ns, {k: '%.3g' % x for k, x in v.items()})
        if ns and 'SQ_WAVE_CYCLES' in v:
            print('   mean resident waves per CU ~', v['SQ_WAVE_CYCLES'] * 4 / (ns * 2.4 * 256))
PY
