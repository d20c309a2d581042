#!/bin/bash
# the front of the first part alone (upload, sketch, index, seed plan), per-kernel HIP-event times, for several builds / knobs:
#   VARIANTS="base|LQCOV_LIBRARY=longqc_amd/var/lib_x.so|LQCOV_SEED_BUCKET=4096" CFG=cfg3 STEPS=2 bash tools/gpu_front.sh
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/front.log
IFS='|' read -ra VS <<< "${VARIANTS:-base}"
for V in "${VS[@]}"; do
  E="$V"; [ "$V" = base ] && E=""
  env $E timeout ${LIMIT:-300} python bench.py --config ${CFG:-cfg3} --front-only --steps ${STEPS:-2} --cache /tmp/lqcov_cache 2>gpurun_out/front_err.log | tail -1 | python -c "
import sys, json
try:
    j = json.loads(sys.stdin.read())
    st = j['stages_ms']
    print('%-60s wall %7.1f  %s' % ('$V', j['wall_ms'], {k: v for k, v in st.items() if k.startswith('k_seed') or k in ('$ALSO'.split(','))}))
except Exception as e:
    print('%-60s failed: %r' % ('$V', e)); print(open('gpurun_out/front_err.log').read()[-600:])
" >> gpurun_out/front.log 2>&1
done
cat gpurun_out/front.log
