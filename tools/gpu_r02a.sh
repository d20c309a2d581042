#!/bin/bash
# round 2, first contact: GPU suite, bench on configs[2] (cfg3) with the packed-upload clock, rocprof kernel stats
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
T0=$(date +%s)
( timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/pytest_gpu.log 2>&1
echo "pytest done $(( $(date +%s) - T0 )) s" >> gpurun_out/pytest_gpu.log
( timeout 400 python bench.py --reads 100000 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -3 ) > gpurun_out/bench_cfg3_100k.log 2>&1
echo "bench100k done $(( $(date +%s) - T0 )) s" >> gpurun_out/bench_cfg3_100k.log
( timeout 900 python bench.py --steps 2 --warmup 1 2>&1 | tail -5 ) > gpurun_out/bench_cfg3.log 2>&1
echo "bench done $(( $(date +%s) - T0 )) s" >> gpurun_out/bench_cfg3.log
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline"
( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o b -- $B 2>&1 | tail -3 ) > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1
cp /tmp/prof/b_kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/ 2>/dev/null
echo "all done $(( $(date +%s) - T0 )) s" >> $GRAFT_REPO_ROOT/gpurun_out/rocprof.log
nproc >> $GRAFT_REPO_ROOT/gpurun_out/rocprof.log; free -g | head -2 >> $GRAFT_REPO_ROOT/gpurun_out/rocprof.log
cd $GRAFT_REPO_ROOT; for f in pytest_gpu bench_cfg3_100k bench_cfg3 rocprof; do echo "== $f"; tail -n 4 gpurun_out/$f.log | cut -c1-3000; done
