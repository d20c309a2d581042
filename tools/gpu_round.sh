#!/bin/bash
# one GPU session: pytest -m gpu, smoke, bench (N=1), rocprof kernel-trace summary -> gpurun_out/
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
T0=$(date +%s)
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/pytest_gpu.log 2>&1
echo "pytest done $(( $(date +%s) - T0 )) s" >> gpurun_out/pytest_gpu.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 ) > gpurun_out/smoke.log 2>&1
( timeout 600 python bench.py --steps 3 --warmup 1 2>&1 | tail -5 ) > gpurun_out/bench.log 2>&1
echo "bench done $(( $(date +%s) - T0 )) s" >> gpurun_out/bench.log
cd /tmp && export TMPDIR=/tmp
( timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --reads 20000 --nsample 2000 --no-cpu-baseline 2>&1 | tail -3 ) > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1
find /tmp/prof -name "*stats*.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/ \; 2>/dev/null
ls -la /tmp/prof/* >> $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1
cd $GRAFT_REPO_ROOT; tail -3 gpurun_out/pytest_gpu.log gpurun_out/smoke.log gpurun_out/bench.log
