#!/bin/bash
# one evidence round: pytest -m gpu, smoke(), bench on configs[2] (default) and configs[1], rocprofv3 --kernel-trace --stats of
# the bench command, and (PMC=1) the counter passes of tools/gpu_pmc.sh.  Every step under its own time limit; the reads of
# configs[2] are generated once (bench.py --cache).  Copy what is to be judged from gpurun_out/ into profiles/ right away:
# tools/gpu.sh wipes gpurun_out/ before a run.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
T0=$(date +%s)
timeout 400 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_full.log 2>&1; tail -3 gpurun_out/pytest_full.log > gpurun_out/pytest_gpu.log
echo "pytest done $(( $(date +%s) - T0 )) s" >> gpurun_out/pytest_gpu.log
( timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 ) > gpurun_out/smoke.log 2>&1
C="--cache /tmp/lqcov_cache"
( timeout 300 python bench.py --steps 3 --warmup 1 $C 2>&1 | tail -1 ) > gpurun_out/bench_cfg3.json
echo "bench done $(( $(date +%s) - T0 )) s" >> gpurun_out/pytest_gpu.log
( timeout 150 python bench.py --config cfg2 --steps 3 --warmup 1 --no-cpu-baseline $C 2>&1 | tail -1 ) > gpurun_out/bench_cfg2.json
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline $C"
( timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o b -- $B 2>&1 | tail -1 ) > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1
cp /tmp/prof/b_kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/kernel_stats_bench_cfg3.csv 2>/dev/null
cd $GRAFT_REPO_ROOT
[ "$PMC" = 1 ] && bash tools/gpu_pmc.sh > gpurun_out/pmc_round.log 2>&1
echo "all done $(( $(date +%s) - T0 )) s" >> gpurun_out/pytest_gpu.log
cat gpurun_out/pytest_gpu.log gpurun_out/smoke.log; head -c 600 gpurun_out/bench_cfg3.json; echo; head -c 300 gpurun_out/bench_cfg2.json; echo
