#!/bin/bash
# the engine's own timeline of one timed step (LQCOV_TIMELINE=1: host times at the points where a thread has waited for its stream),
# no profiler in the way:   [ENVV="LQCOV_LANES=4"] bash tools/gpu_timeline.sh   -> gpurun_out/timeline_host.txt (the last step)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
env LQCOV_TIMELINE=1 $ENVV timeout ${LIMIT:-300} python bench.py --config ${CFG:-cfg3} --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --no-north-star --cache /tmp/lqcov_cache 2> gpurun_out/timeline_err.log | tail -1 > gpurun_out/timeline_bench.json
python - <<'PY'
import json
lines = [l for l in open("gpurun_out/timeline_err.log") if l.startswith("[tl]")]
starts = [i for i, l in enumerate(lines) if l.split()[2] == "main0" and "reset" in l]
# steps: warm-up, the profiled one, two timed ones; the last-but-one reset opens a timed step
a = starts[-2] if len(starts) >= 2 else starts[-1]
b = starts[-1]
open("gpurun_out/timeline_host.txt", "w").writelines(lines[a:b])
print("".join(lines[a:b]))
try:
    j = json.loads(open("gpurun_out/timeline_bench.json").read())
    print(j["value"], j["ms_per_step"])
except Exception as e:
    print("no bench line:", e)
PY
