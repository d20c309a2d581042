#!/bin/bash
# round 2: GPU suite + bench on configs[2] and configs[1]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
T0=$(date +%s)
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/pytest_gpu.log 2>&1
echo "pytest done $(( $(date +%s) - T0 )) s" >> gpurun_out/pytest_gpu.log
( timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -3 ) > gpurun_out/bench_cfg3.log 2>&1
echo "bench done $(( $(date +%s) - T0 )) s" >> gpurun_out/bench_cfg3.log
( timeout 600 python bench.py --config cfg2 --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | tail -3 ) > gpurun_out/bench_cfg2.log 2>&1
echo "bench2 done $(( $(date +%s) - T0 )) s" >> gpurun_out/bench_cfg2.log
for f in pytest_gpu bench_cfg3 bench_cfg2; do echo "== $f"; tail -n 4 gpurun_out/$f.log | cut -c1-3500; done
