#!/bin/bash
# one evidence round: pytest -m gpu, smoke(), bench on configs[2] (default; CPUFULL=1: with the reference timed on ALL reads) and
# configs[1], rocprofv3 --kernel-trace --stats of the bench command, and (PMC=1) two counter passes (FETCH_SIZE, WRITE_SIZE) over
# the full configs[2] step restricted to the kernels named in PMC_RE.  Every step under its own time limit; the reads of
# configs[2] are generated once (bench.py --cache).  Copy what is to be judged from gpurun_out/ into profiles/ right away.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
T0=$(date +%s)
if [ "$SKIP_TESTS" != 1 ]; then
  timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_full.log 2>&1; tail -3 gpurun_out/pytest_full.log > gpurun_out/pytest_gpu.log
  echo "pytest done $(( $(date +%s) - T0 )) s" >> gpurun_out/pytest_gpu.log
  ( timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 ) > gpurun_out/smoke.log 2>&1
fi
C="--cache /tmp/lqcov_cache"
CPU=""; [ "$CPUFULL" = 1 ] && CPU="--cpu-sample 500000"
( timeout ${BENCH_LIMIT:-900} python bench.py --steps ${STEPS:-5} --warmup 2 $C $CPU $BENCH_EXTRA 2>gpurun_out/bench_err.log | tail -1 ) > gpurun_out/bench_cfg3.json   # BENCH_EXTRA="--gz-end-to-end": the files-in / table-out call once more on a .gz
echo "bench done $(( $(date +%s) - T0 )) s" >> gpurun_out/pytest_gpu.log
( timeout 150 python bench.py --config cfg2 --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end $C 2>&1 | tail -1 ) > gpurun_out/bench_cfg2.json
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end --no-north-star $C"
( timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o b -- $B 2>&1 | tail -1 ) > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1
cp /tmp/prof/b_kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/kernel_stats_bench_cfg3.csv 2>/dev/null
echo "rocprof done $(( $(date +%s) - T0 )) s" >> $GRAFT_REPO_ROOT/gpurun_out/pytest_gpu.log
if [ "$PMC" = 1 ]; then
  R=$GRAFT_REPO_ROOT
  RE=${PMC_RE:-'k_seed_count|k_seed_scatter|k_seed_decide|k_seed_emit|k_ps_finish|k_ps_scatter|k_ps_hist|k_chain|k_run_list|k_rs_scatter|k_rs_hist|k_sort_walk_solo|k_ck_chain256|k_sketch|k_is_pass|k_is_hist|k_head_lookback'}
  BP="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end --no-north-star $C"
  for CN in FETCH_SIZE WRITE_SIZE; do
    ( timeout ${PMC_LIMIT:-300} rocprofv3 --pmc $CN --kernel-trace --kernel-include-regex "$RE" --output-format csv -d /tmp/pmc_$CN -o p -- $BP 2>&1 | tail -3 ) > $R/gpurun_out/pmc_$CN.log 2>&1
    echo "pmc $CN done $(( $(date +%s) - T0 )) s" >> $R/gpurun_out/pytest_gpu.log
  done
  python $R/tools/pmc_to_json.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE $R/gpurun_out/pmc_traffic.json "bench.py --config cfg3 (configs[2]: 500000 reads, 5000 queries), default lanes, 1 step; counters for: $RE" > $R/gpurun_out/pmc_summary.txt 2>&1
fi
cd $GRAFT_REPO_ROOT
echo "all done $(( $(date +%s) - T0 )) s" >> gpurun_out/pytest_gpu.log
cat gpurun_out/pytest_gpu.log gpurun_out/smoke.log; head -c 900 gpurun_out/bench_cfg3.json; echo; head -c 300 gpurun_out/bench_cfg2.json; echo; cat gpurun_out/pmc_summary.txt 2>/dev/null
