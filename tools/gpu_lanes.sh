#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for L in 2 4 6; do
  LQCOV_LANES=$L timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('lanes $L', j['value'], j['ms_per_step'], (j.get('golden_rows') or {}).get('rows_identical'), {k: round(v) for k, v in list(j['roofline']['kernel_ms_one_step'].items())[:8]})
" >> gpurun_out/lanes.log 2>&1
done
cat gpurun_out/lanes.log
