#!/bin/bash
# one GPU session: pytest -m gpu, smoke, bench (N=1), rocprof kernel-trace summary, PMC passes -> gpurun_out/
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
T0=$(date +%s)
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/pytest_gpu.log 2>&1
echo "pytest done $(( $(date +%s) - T0 )) s" >> gpurun_out/pytest_gpu.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 ) > gpurun_out/smoke.log 2>&1
( timeout 900 python bench.py --steps 3 --warmup 1 2>&1 | tail -5 ) > gpurun_out/bench.log 2>&1
echo "bench done $(( $(date +%s) - T0 )) s" >> gpurun_out/bench.log
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline"
( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o b -- $B 2>&1 | tail -3 ) > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1
cp /tmp/prof/b_kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/ 2>/dev/null
for C in FETCH_SIZE WRITE_SIZE; do
  ( timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_$C -o p -- $B 2>&1 | tail -3 ) > $GRAFT_REPO_ROOT/gpurun_out/pmc_$C.log 2>&1
done
python $GRAFT_REPO_ROOT/tools/pmc_to_json.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE $GRAFT_REPO_ROOT/gpurun_out/pmc_traffic.json "bench.py default (configs[1]: 50000 reads, 5000 queries), 4 steps" > $GRAFT_REPO_ROOT/gpurun_out/pmc_summary.txt 2>&1
cd $GRAFT_REPO_ROOT; for f in pytest_gpu smoke bench; do tail -n 3 gpurun_out/$f.log | cut -c1-600; done; head -n 20 gpurun_out/pmc_summary.txt
