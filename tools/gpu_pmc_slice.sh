#!/bin/bash
# HBM traffic of every kernel on a slice of the bench workload that finishes under counter collection (every dispatch is
# serialised): one rocprofv3 --pmc pass per counter (FETCH_SIZE and WRITE_SIZE do not fit one pass), one mapping lane.
#   READS=100000 NSAMPLE=1000 bash tools/gpu_pmc_slice.sh   ->  gpurun_out/pmc_slice_traffic.json + pmc_slice_summary.txt
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
READS=${READS:-100000}; NS=${NSAMPLE:-1000}
B="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --reads $READS --nsample $NS --cache /tmp/lqcov_cache"
LQCOV_LANES=1 timeout 200 $B > $R/gpurun_out/pmc_slice_plain.json 2> $R/gpurun_out/pmc_slice_plain.err
for C in FETCH_SIZE WRITE_SIZE; do
  ( LQCOV_LANES=1 timeout ${PMC_LIMIT:-300} rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmcs_$C -o p -- $B 2>&1 | tail -3 ) > $R/gpurun_out/pmc_slice_$C.log 2>&1
done
python $R/tools/pmc_to_json.py /tmp/pmcs_FETCH_SIZE /tmp/pmcs_WRITE_SIZE $R/gpurun_out/pmc_slice_traffic.json "bench.py --config cfg3 --reads $READS --nsample $NS (a slice of configs[2]), LQCOV_LANES=1, 2 passes, every kernel" > $R/gpurun_out/pmc_slice_summary.txt 2>&1
cat $R/gpurun_out/pmc_slice_summary.txt; head -c 1500 $R/gpurun_out/pmc_slice_plain.json
