#!/bin/bash
# Round-3 evidence in one gpurun call: rocprofv3 summaries of the bench (kernel-trace + separate --pmc passes), lane
# occupancy of the one-tile and the 4 096-chain launches, phase clocks of the deep-round kernel (diagnostic build), the
# default bench line, the dataset block with two ranks (gloo, one GPU), the contrast-setup chains.
TAG=${1:-r03}
export TMPDIR=/tmp
mkdir -p gpurun_out
bash profiles/run_profile.sh $TAG > gpurun_out/${TAG}_run_profile.log 2>&1
python3 profiles/tools/summarize.py $TAG > /dev/null 2>&1; cp profiles/${TAG}_summary.md gpurun_out/ 2>/dev/null
echo "run_profile done"
bash profiles/tools/pmc_lanes.sh $TAG > gpurun_out/${TAG}_lanes.md 2> gpurun_out/${TAG}_lanes.err
echo "pmc_lanes done"
MPP_LIB_PATH=$PWD/mpp_cnn_rs_object_detection_amd/libmppgpu_dprof.so python3 profiles/tools/deep_probe.py --reps 1 --configs 8:128:0 > gpurun_out/${TAG}_phases.json 2>/dev/null
echo "phases done"
python3 bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
echo "bench done"
MPP_DIST_BACKEND=gloo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-convergence --batched-tiles 0 --scene 0 --mosaic 0 > gpurun_out/${TAG}_dataset_2ranks.json 2> gpurun_out/${TAG}_dataset_2ranks.err
echo "2-rank dataset done"
tail -c 400 gpurun_out/${TAG}_dataset_2ranks.json
