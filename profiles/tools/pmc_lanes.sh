#!/bin/bash
# Lane occupancy of the chain kernel: SQ_THREAD_CYCLES_VALU beside SQ_ACTIVE_INST_VALU
# (active lanes per VALU cycle) for the one-tile launch and the many-chains launch, each its own --pmc pass with
# --kernel-trace only.   bash profiles/tools/pmc_lanes.sh <tag> [extra bench.py args for the batched launch]
export TMPDIR=/tmp
TAG=$1; shift
OUT=$PWD/gpurun_out/prof_lanes_$TAG
mkdir -p $OUT
CTRS="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VALU SQ_INSTS_SALU"
ONE="--steps 2 --warmup 1 --no-cpu-baseline --no-convergence --batched-tiles 0 --scene 0 --mosaic 0 --dataset-images 0"
MANY="--steps 1 --warmup 0 --no-cpu-baseline --no-convergence --scene 0 --mosaic 0 --dataset-images 0 --batched-tiles 4096 $@"
rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT/one -- python3 bench.py $ONE > $OUT/bench_one.json 2> $OUT/err_one.txt
echo "one-tile pass done" >&2
rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT/many -- python3 bench.py $MANY > $OUT/bench_many.json 2> $OUT/err_many.txt
echo "many-chains pass done" >&2
python3 profiles/tools/summarize_lanes.py $TAG | tee $OUT/summary.md
