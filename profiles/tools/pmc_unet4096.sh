#!/bin/bash
# rocprofv3 evidence for the 4096 x 4096 float32 score-map forward (BASELINE config 5): per-kernel time, then MFMA busy
# cycles and HBM bytes in their own --pmc passes (program directly after --, --kernel-trace only)
export TMPDIR=/tmp
TAG=${1:-r03}
OUT=$PWD/gpurun_out/prof_unet4096_$TAG
mkdir -p $OUT
# the forward as the product runs it (PosNet and ShapeNet on a stream each) ...
python3 profiles/tools/prof_unet_4096.py > $OUT/bench_two_streams.json 2>/dev/null
# ... and on ONE stream under the profiler: with two, kernels of the two networks overlap and a kernel's duration says little
export MPP_UNET_TWO_STREAMS=0
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 profiles/tools/prof_unet_4096.py > $OUT/bench.json 2> $OUT/trace.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc1 -- python3 profiles/tools/prof_unet_4096.py > /dev/null 2> $OUT/pmc1.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc2 -- python3 profiles/tools/prof_unet_4096.py > /dev/null 2> $OUT/pmc2.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc3 -- python3 profiles/tools/prof_unet_4096.py > /dev/null 2> $OUT/pmc3.err
python3 profiles/tools/summarize_unet.py $TAG
