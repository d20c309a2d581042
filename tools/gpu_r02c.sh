#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
T0=$(date +%s)
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_full.log 2>&1
tail -4 gpurun_out/pytest_full.log > gpurun_out/pytest_gpu.log
echo "pytest done $(( $(date +%s) - T0 )) s" >> gpurun_out/pytest_gpu.log
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline"
( timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o b -- $B 2>&1 | tail -3 ) > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1
python - <<'PY'
import csv, os, collections
fn = "/tmp/prof/b_kernel_trace.csv"
rows = list(csv.DictReader(open(fn)))
print("kernels", len(rows), rows[0].keys())
out = open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/trace_small.csv", "w")
t0 = min(int(r["Start_Timestamp"]) for r in rows)
for r in rows:
    name = r["Kernel_Name"].split("(")[0][:40]
    out.write("%s,%s,%d,%d\n" % (name, r.get("Stream_Id", r.get("Queue_Id", "")), int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0))
PY
cd $GRAFT_REPO_ROOT; ls -la gpurun_out; cat gpurun_out/pytest_gpu.log; tail -3 gpurun_out/rocprof.log | cut -c1-300
