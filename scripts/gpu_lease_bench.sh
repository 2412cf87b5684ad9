#!/bin/bash
# One fresh lease: the box's clocks / power state, then the DRIVER's bench command (python3 bench.py --gpus 1 --steps 20 --warmup 5) with its wall time.
# usage (through gpurun): bash scripts/gpu_lease_bench.sh <n>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
N=${1:-1}
O=gpurun_out/r6_bench_lease$N
{ rocm-smi --showclocks 2>/dev/null | grep -iE "sclk|mclk|fclk|socclk" | head -4; rocm-smi --showpower 2>/dev/null | grep -i "power" | head -1; rocm-smi --showperflevel 2>/dev/null | grep -i "perf" | head -1; } > $O.box 2>&1
T0=$(date +%s%N)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O.json 2> $O.err
T1=$(date +%s%N)
echo "driver_run_s $(( (T1 - T0) / 1000000 )) ms" >> $O.box
python - <<PY
import json
d = json.load(open("$O.json"))
r = d["roofline"]
print("lease $N:", round(d["value"], 2), "structures/s", round(d["ms_per_step"], 3), "ms/step frac", round(r["frac"], 3), "avg_launch_ms", round(r["avg_launch_ms"], 4), "traffic MB", None if r["traffic"] is None else round(r["traffic"] / 1e6, 1))
e = d.get("extra", {})
print("  exact fp32", round(e["exact_fp32_path"]["value"], 2), "| tf32-class", e.get("tf32_class_path", {}).get("value"), "| mg", e.get("mattergen_shaped_sampler", {}).get("value"), "| ft", e.get("fine_tune", {}).get("value"), "ft20", e.get("fine_tune", {}).get("value_20step"))
PY
cat $O.box
