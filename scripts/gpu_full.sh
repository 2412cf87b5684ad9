#!/bin/bash
# whole GPU suite + the driver-shaped default line at HEAD
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | tail -25 > gpurun_out/full_pytest.log
tail -22 gpurun_out/full_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 900 python bench.py --steps 20 --warmup 3 > gpurun_out/bench20.json 2> gpurun_out/bench20.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench20.json"))
print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["frac"], d["cpu_baseline"]["value"])
print({k: (v.get("value"), v.get("error")) for k, v in d["extra"].items()})
print(d["extra"]["fine_tune"].get("roofline", {}).get("frac"), d["extra"]["mattergen_shaped_sampler"].get("hbm_roofline"))
PY
tail -3 gpurun_out/bench20.err
