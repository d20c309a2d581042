#!/bin/bash
# Strong-scaling curve of the bench workload (BASELINE configs[2]) on one node: N = 1, 2, 4, 8 ranks, one per GPU, RCCL over
# xGMI (the driver's launch line).  Prints per N: Mbases/s, ms per step, the parallelism the line reports, the number of RCCL
# ranks that took part and the golden-row check.  Needs N GPUs; on a 1-GPU box only N = 1 runs.
#   STEPS=5 WARMUP=2 [CONFIG=cfg4s] bash tools/scale.sh      (prints the stage-rate model's prediction beside the measurement)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
NGPU=$(python -c "import torch; print(torch.cuda.device_count())")
for N in 1 2 4 8; do
  [ "$N" -gt "$NGPU" ] && { echo "N=$N: skipped ($NGPU GPU(s) here)"; continue; }
  if [ "$N" = 1 ]; then L="python bench.py"; else L="python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500 + N)) bench.py"; fi
  HSA_ENABLE_IPC_MODE_LEGACY=0 $L --gpus $N --steps ${STEPS:-5} --warmup ${WARMUP:-2} --no-cpu-baseline --no-end-to-end --no-north-star ${CONFIG:+--config $CONFIG} --cache /tmp/lqcov_cache 2>/tmp/scale_$N.err | tail -1 | python -c "
import sys, json
try:
    j = json.loads(sys.stdin.read())
    m = j['config'].get('scaling_model_s') or {}
    print('N=%d  %9.1f Mbases/s  %8.1f ms/step  n_gpus=%s  rows %s/%s  model (s per job): %s  %s' % ($N, j['value'], j['ms_per_step'], j['n_gpus'],
          (j.get('golden_rows') or {}).get('rows_identical'), (j.get('golden_rows') or {}).get('rows_checked'),
          {k: v for k, v in m.items() if k != 'note'} or '-', j['config']['parallelism'][:80]))
except Exception as e:
    print('N=$N failed: %r' % e); print(open('/tmp/scale_$N.err').read()[-800:])
"
done
