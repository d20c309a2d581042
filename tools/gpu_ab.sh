#!/bin/bash
# A/B of engine knobs on the bench workload in ONE gpurun call: the reads are generated once (bench.py --cache), every
# variant is one bench run under its own time limit; one line per variant in gpurun_out/ab.log.
#   VARIANTS="base|LQCOV_LANES=1|LQCOV_CKPT3=1|LQCOV_LANES=3 LQCOV_CKPT3=1"  CFG=cfg3  STEPS=3  bash tools/gpu_ab.sh
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/ab.log
# PYTEST_K="randomised or switch": a parity subset first
if [ -n "$PYTEST_K" ]; then env $PYTEST_ENV timeout ${PYTEST_LIMIT:-120} python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$PYTEST_K" 2>&1 | tail -3 >> gpurun_out/ab.log; fi
IFS='|' read -ra VS <<< "${VARIANTS:-base|LQCOV_LANES=1|LQCOV_LANES=2}"
N=0
for V in "${VS[@]}"; do
  E="$V"; [ "$V" = base ] && E=""
  N=$((N+1))
  env $E timeout ${LIMIT:-300} python bench.py --config ${CFG:-cfg3} --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline --no-end-to-end --no-north-star --cache /tmp/lqcov_cache 2>gpurun_out/ab_err.log | tail -1 | tee gpurun_out/ab_$N.json | python -c "
import sys, json
try:
    j = json.loads(sys.stdin.read())
    print('%-40s %9.1f Mbases/s %8.1f ms  rows %s  %s' % ('$V', j['value'], j['ms_per_step'], (j.get('golden_rows') or {}).get('rows_identical'),
          {k: round(v) for k, v in list(j['roofline']['kernel_ms_one_step'].items())[:14]}))
except Exception as e:
    print('%-40s failed: %r' % ('$V', e)); print(open('gpurun_out/ab_err.log').read()[-600:])
" >> gpurun_out/ab.log 2>&1
done
cat gpurun_out/ab.log
