#!/bin/bash
# kernel trace + FETCH_SIZE / WRITE_SIZE passes of the MatterGen-shaped sampler line, summarised on the box
TAG=${2:-r3}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python bench.py --mode mg-sample --steps 3 --warmup 1 --mg-chains 1 --no-cpu-baseline"   # (one chain: kernels run one after the other, so a kernel's bytes / its duration is ITS rate; with concurrent chains the durations overlap)
rocprofv3 --kernel-trace --stats -d gpurun_out/pm_t -o mg -- $CMD > gpurun_out/pm_t.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pm_f -o mg -- $CMD > gpurun_out/pm_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pm_w -o mg -- $CMD > gpurun_out/pm_w.log 2>&1
python scripts/rocprof_summary.py gpurun_out/${TAG}_rocprofv3_summary_mattergen_sampler.md gpurun_out/pm_t/mg_results.db gpurun_out/pm_f/mg_results.db gpurun_out/pm_w/mg_results.db > /dev/null
python - <<PY
import json
p = "gpurun_out/${TAG}_rocprofv3_summary_mattergen_sampler_traffic.json"
d = json.load(open(p)); d["head"] = "${1:-unknown}"; d["command"] = "$CMD"; d["steps_in_trace"] = 4
json.dump(d, open(p, "w"), indent=1)
PY
rm -rf gpurun_out/pm_t gpurun_out/pm_f gpurun_out/pm_w
grep -n "HBM-side traffic per" -A20 gpurun_out/${TAG}_rocprofv3_summary_mattergen_sampler.md | cut -c1-130
