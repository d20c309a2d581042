cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
LQCOV_BENCH_ONE_DEVICE=1 timeout 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 2 --warmup 1 --reads 8000 --nsample 800 2>&1 | tail -5 | tee gpurun_out/bench_2rank_onedev.log
timeout 300 python bench.py --steps 2 --warmup 1 --reads 8000 --nsample 800 --no-cpu-baseline 2>&1 | tail -2 | tee gpurun_out/bench_1rank_small.log
