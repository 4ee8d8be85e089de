export MPP_DIST_BACKEND=gloo
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/r02_bench_2rank.json 2> gpurun_out/r02_bench_2rank.err
tail -c 600 gpurun_out/r02_bench_2rank.json; tail -3 gpurun_out/r02_bench_2rank.err
