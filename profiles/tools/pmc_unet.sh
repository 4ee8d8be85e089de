#!/bin/bash
# rocprofv3 evidence for the score-map forward: per-kernel time, MFMA busy cycles, HBM bytes (separate --pmc passes)
set -o pipefail
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_unet
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 profiles/tools/bench_unet.py > $OUT/bench.json 2> $OUT/trace.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $OUT/pmc1 -- python3 profiles/tools/bench_unet.py > /dev/null 2> $OUT/pmc1.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc2 -- python3 profiles/tools/bench_unet.py > /dev/null 2> $OUT/pmc2.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc3 -- python3 profiles/tools/bench_unet.py > /dev/null 2> $OUT/pmc3.err
ls $OUT/*/*/ | head -30
