#!/bin/bash
# one full evidence round (round 2): pytest -m gpu, smoke(), bench on configs[2] (default) and configs[1], rocprofv3
# --kernel-trace --stats, two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), the N>1 path on one device over gloo
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
T0=$(date +%s)
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_full.log 2>&1; tail -3 gpurun_out/pytest_full.log > gpurun_out/pytest_gpu.log
echo "pytest done $(( $(date +%s) - T0 )) s" >> gpurun_out/pytest_gpu.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ) > gpurun_out/smoke.log 2>&1
( timeout 600 python bench.py --steps 5 --warmup 1 2>&1 | tail -1 ) > gpurun_out/bench_cfg3.json 2> gpurun_out/bench_cfg3.err
echo "bench done $(( $(date +%s) - T0 )) s" >> gpurun_out/pytest_gpu.log
( timeout 300 python bench.py --config cfg2 --steps 5 --warmup 1 2>&1 | tail -1 ) > gpurun_out/bench_cfg2.json
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline"
( timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o b -- $B 2>&1 | tail -2 ) > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1
cp /tmp/prof/b_kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/kernel_stats_bench_cfg3.csv 2>/dev/null
B1="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline"
for C in FETCH_SIZE WRITE_SIZE; do
  ( timeout 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_$C -o p -- $B1 2>&1 | tail -2 ) > $GRAFT_REPO_ROOT/gpurun_out/pmc_$C.log 2>&1
done
python $GRAFT_REPO_ROOT/tools/pmc_to_json.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE $GRAFT_REPO_ROOT/gpurun_out/pmc_traffic_cfg3.json "bench.py default (cfg3 = BASELINE configs[2]: 500000 reads, 5000 queries), 2 steps (1 profiled + 1 timed)" > $GRAFT_REPO_ROOT/gpurun_out/pmc_summary.txt 2>&1
cd $GRAFT_REPO_ROOT
( LQCOV_BENCH_ONE_DEVICE=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 2 --reads 100000 --steps 2 --warmup 1 2>&1 | grep '^{' | tail -1 ) > gpurun_out/bench_2ranks_one_device.json
echo "all done $(( $(date +%s) - T0 )) s" >> gpurun_out/pytest_gpu.log
cat gpurun_out/pytest_gpu.log gpurun_out/smoke.log; head -c 900 gpurun_out/bench_cfg3.json; echo; head -c 300 gpurun_out/bench_cfg2.json; echo; head -n 14 gpurun_out/pmc_summary.txt; head -c 400 gpurun_out/bench_2ranks_one_device.json
