#!/bin/bash
# Profile recipe (run on the GPU box through gpurun):  bash profiles/run_profile.sh <tag> [bench args...]
# 1. rocprofv3 --kernel-trace --stats  -> per-kernel durations
# 2. separate --pmc passes (never combined with trace domains other than kernel-trace)
set -o pipefail
TAG=${1:-r01}; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-convergence --batched-tiles 0 --scene 0 --mosaic 0 --dataset-images 0 $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d $OUT/pmc1 -- python3 bench.py $ARGS > /dev/null 2> $OUT/pmc1.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc2 -- python3 bench.py $ARGS > /dev/null 2> $OUT/pmc2.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -- python3 bench.py $ARGS > /dev/null 2> $OUT/pmc3.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc4 -- python3 bench.py $ARGS > /dev/null 2> $OUT/pmc4.err
find $OUT -name "*.csv" | head -50
