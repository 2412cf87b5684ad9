#!/bin/bash
# every secondary bench line at one HEAD -> gpurun_out/r5_bench_<mode>.json
cd $GRAFT_REPO_ROOT
for mode in mg-sample sample-default ft-default mg-ft; do
  python bench.py --mode $mode $([ $mode = mg-ft ] && echo --mg-batch 256) 2> gpurun_out/r5_bench_$mode.err | tail -1 > gpurun_out/r5_bench_$mode.json
  python - <<PY
import json
d = json.load(open("gpurun_out/r5_bench_$mode.json"))
r = d.get("roofline") or d.get("hbm_roofline") or {}
print("$mode", round(d["value"], 3), d["unit"], "frac", r.get("frac"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
PY
done
