#!/bin/bash
# Is one bench process using the whole device?  The same bench (configs[1], fixed batch size) alone, then two copies side by side:
# if the two take about as long as the one, the device had room and the limit is on the host side of a process.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export LQCOV_ANCHOR_BUDGET=${BUDGET:-100000000}
B="python bench.py --config cfg2 --steps 6 --warmup 2 --no-cpu-baseline --cache /tmp/lqcov_cache"
$B 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('one process:   %.1f ms per step' % j['ms_per_step'])" | tee gpurun_out/two_procs.log
( $B 2>/dev/null | tail -1 > /tmp/p1.json ) & ( $B 2>/dev/null | tail -1 > /tmp/p2.json ) & wait
python -c "
import json
for f in ('/tmp/p1.json', '/tmp/p2.json'):
    j = json.loads(open(f).read()); print('two processes: %.1f ms per step' % j['ms_per_step'])
" | tee -a gpurun_out/two_procs.log
