#!/bin/bash
# bench with the north-star pass (every kernel alone on the device): step time + the seed / sketch kernels' own times
#   VARIANTS="base|LQCOV_SEED_BUCKET=4096" CFG=cfg3 STEPS=2 bash tools/gpu_north.sh
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/north.log
IFS='|' read -ra VS <<< "${VARIANTS:-base}"
N=0
for V in "${VS[@]}"; do
  E="$V"; [ "$V" = base ] && E=""
  N=$((N+1))
  env $E timeout ${LIMIT:-400} python bench.py --config ${CFG:-cfg3} --steps ${STEPS:-2} --warmup 1 --no-cpu-baseline --no-end-to-end --cache /tmp/lqcov_cache 2>gpurun_out/north_err.log | tail -1 > gpurun_out/north_$N.json
  python - "$V" gpurun_out/north_$N.json >> gpurun_out/north.log 2>&1 <<'PY'
import sys, json
try:
    j = json.load(open(sys.argv[2]))
    n = j.get("north_star") or j["roofline"].get("north_star") or {}
    print("%-44s %8.1f Mbases/s %7.1f ms rows %s written %s" % (sys.argv[1], j["value"], j["ms_per_step"], (j.get("golden_rows") or {}).get("rows_identical"), j["config"].get("anchors_written_per_step")))
    if n:
        print("   seed alone:", n["seed"]["kernels"], n["seed"]["ms"], "frac", n["seed"]["frac_of_hbm_peak"])
        print("   sketch alone:", n["sketch"]["kernels"], n["sketch"]["ms"], "frac", n["sketch"]["frac_of_hbm_peak"], "combined", n["combined"]["frac_of_hbm_peak"])
    print("   one step:", {k: round(v) for k, v in list(j["roofline"]["kernel_ms_one_step"].items())[:18]})
except Exception as e:
    print("%-44s failed: %r" % (sys.argv[1], e)); print(open("gpurun_out/north_err.log").read()[-800:])
PY
done
cat gpurun_out/north.log
