#!/bin/bash
# issue / occupancy / HBM counters of the many-chains launch (4096 tiles, one wave per chain), separate passes
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_batched_$1; shift
mkdir -p $OUT
ARGS="--steps 1 --warmup 0 --no-cpu-baseline --no-convergence --batched-tiles 4096 $@"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/p1 -- python3 bench.py $ARGS > $OUT/bench.json 2> $OUT/err1.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/p2 -- python3 bench.py $ARGS > /dev/null 2> $OUT/err2.txt
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/p3 -- python3 bench.py $ARGS > /dev/null 2> $OUT/err3.txt
python3 - <<PY
import csv, glob, collections
best = {}
for p in sorted(glob.glob("$OUT/p*/*/*_counter_collection.csv")):
    kt = p.replace("_counter_collection.csv", "_kernel_trace.csv")
    dur = {r['Dispatch_Id']: int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in csv.DictReader(open(kt)) if 'mpp_chain' in r['Kernel_Name']}
    agg = collections.defaultdict(dict)
    for r in csv.DictReader(open(p)):
        if 'mpp_chain' in r['Kernel_Name'] and '<1, 0, false, 2' in r['Kernel_Name']: agg[r['Dispatch_Id']][r['Counter_Name']] = float(r['Counter_Value'])
    if not agg: continue
    d = max(agg, key=lambda k: dur.get(k, 0))                 # the 30 257-step launch (not its 2 000-step warm-up)
    ns = dur[d]
    print('| launch (ms) | %.1f |' % (ns / 1e6))
    for k, x in agg[d].items(): print('| %s | %.4g |' % (k, x))
    v = agg[d]
    if 'SQ_ACTIVE_INST_VALU' in v: print('| VALU busy per SIMD, all 256 CUs (ACTIVE_INST_VALU x4 / 1024 SIMDs / cycles at 2.4 GHz) | %.3f |' % (v['SQ_ACTIVE_INST_VALU'] * 4 / 1024 / (ns * 2.4)))
    if 'SQ_WAVE_CYCLES' in v: print('| mean resident waves per CU (WAVE_CYCLES x4 / cycles / 256) | %.2f |' % (v['SQ_WAVE_CYCLES'] * 4 / (ns * 2.4 * 256)))
    if 'FETCH_SIZE' in v: print('| HBM read GB/s (FETCH_SIZE KiB x2 gfx950 correction) | %.1f |' % (v['FETCH_SIZE'] * 2 * 1024 / ns))
    if 'WRITE_SIZE' in v: print('| HBM write GB/s | %.1f |' % (v['WRITE_SIZE'] * 1024 / ns))
PY
tail -2 $OUT/err1.txt
