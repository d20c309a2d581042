#!/bin/bash
# counters of one kernel (or a few) inside the bench step: one rocprofv3 --pmc pass per counter set (sets separated by '|'),
# restricted to the kernels matching RE; a table of per-launch sums in gpurun_out/pmc_kernel.txt.
#   RE='k_seed_count' SETS='FETCH_SIZE SQ_WAVE_CYCLES SQ_WAIT_ANY|TCC_HIT_sum TCC_MISS_sum' bash tools/gpu_pmc_kernel.sh
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
C="--cache /tmp/lqcov_cache"
python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end --no-north-star $C ${BENCH_ARGS} > /dev/null 2>&1      # (generates and caches the reads)
cd /tmp && export TMPDIR=/tmp
: > $R/gpurun_out/pmc_kernel.txt
IFS='|' read -ra PS <<< "${SETS:-FETCH_SIZE|WRITE_SIZE}"
N=0
for S in "${PS[@]}"; do
  N=$((N+1))
  ( env $ENVV timeout ${LIMIT:-300} rocprofv3 --pmc $S --kernel-trace --kernel-include-regex "${RE:-k_seed_count}" --output-format csv -d /tmp/pmck_$N -o p -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end --no-north-star $C ${BENCH_ARGS} 2>&1 | tail -2 ) > $R/gpurun_out/pmc_kernel_$N.log 2>&1
  python - "$N" "$S" >> $R/gpurun_out/pmc_kernel.txt <<'PY'
import sys, glob, csv, collections, os
n, s = sys.argv[1], sys.argv[2]
f = glob.glob("/tmp/pmck_%s/**/*counter_collection.csv" % n, recursive=True)
agg = collections.defaultdict(lambda: [0.0, 0])
if f:
    for row in csv.DictReader(open(f[0])):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        a = agg[(k, row["Counter_Name"])]; a[0] += float(row["Counter_Value"]); a[1] += 1
print("pass %s: %s" % (n, s))
for (k, c), (v, cnt) in sorted(agg.items()):
    print("  %-34s %-28s launches %-4d sum %.6g  per launch %.6g" % (k[:34], c, cnt, v, v / max(cnt, 1)))
PY
done
cat $R/gpurun_out/pmc_kernel.txt
