#!/bin/bash
# the files-in / table-out call alone (bench.py's end_to_end leg) with run_files' own stage times: gpurun_out/e2e.json
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for V in "${VARIANTS:-base}"; do :; done
IFS='|' read -ra VS <<< "${VARIANTS:-base}"
for V in "${VS[@]}"; do
  E="$V"; [ "$V" = base ] && E=""
  env $E timeout 600 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --cache /tmp/lqcov_cache 2>gpurun_out/e2e_err.log | tail -1 > gpurun_out/e2e.json
  python -c "
import json; j=json.loads(open('gpurun_out/e2e.json').read()); e=j['end_to_end']
print('$V', j['value'], j['ms_per_step'], e['value'], e['seconds'], e['table_identical_to_timed_steps']); print('\n'.join(e['log_tail']))"
done
