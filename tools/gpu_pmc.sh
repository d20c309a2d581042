#!/bin/bash
# HBM traffic of the sort's streaming kernels on the bench workload: one rocprofv3 --pmc pass per counter (counters only for
# the named kernels, so the other ~8000 dispatches of a step run at speed), each under its own time limit.
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
B="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --cache /tmp/lqcov_cache"
# (a plain run first: it pages torch in and fills the cache, so that the counter passes spend their limit on the step itself)
timeout 200 $B > $R/gpurun_out/pmc_plain.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  ( timeout ${PMC_LIMIT:-240} rocprofv3 --pmc $C --kernel-trace --kernel-include-regex 'k_sort_scatter_tiled|k_sort_copy_hist_tiled' --output-format csv -d /tmp/pmc_$C -o p -- $B 2>&1 | tail -3 ) > $R/gpurun_out/pmc_$C.log 2>&1
done
python $R/tools/pmc_to_json.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE $R/gpurun_out/pmc_traffic.json "bench.py --config cfg3 (configs[2]: 500000 reads, 5000 queries), 2 passes; counters for k_sort_scatter_tiled, k_sort_copy_hist_tiled only" > $R/gpurun_out/pmc_summary.txt 2>&1
cat $R/gpurun_out/pmc_summary.txt; tail -n 2 $R/gpurun_out/pmc_*.log | cut -c1-300
