#!/bin/bash
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --cache /tmp/lqcov_cache"
( timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o b -- $B 2>&1 | tail -2 ) > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1
python - <<'PY'
import csv, os
rows = list(csv.DictReader(open("/tmp/prof/b_kernel_trace.csv")))
out = open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/trace_small.csv", "w")
t0 = min(int(r["Start_Timestamp"]) for r in rows)
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace(",", ";")[:48]
    out.write("%s,%s,%d,%d,%s\n" % (name, r["Stream_Id"], int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, r["Grid_Size_X"]))
PY
