#!/bin/bash
# bench on configs[2] (and optionally configs[1]) only, compact output
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for C in ${CFGS:-cfg3}; do
  timeout 400 python bench.py --config $C --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_$C.json
  python -c "
import sys, json
j = json.loads(open('gpurun_out/bench_$C.json').read())
print('$C', j['value'], j['ms_per_step'], (j.get('golden_rows') or {}).get('rows_identical'), j['roofline']['kernel'], j['roofline']['frac'])
print({k: round(v) for k, v in list(j['roofline']['kernel_ms_one_step'].items())[:16]})
"
done
