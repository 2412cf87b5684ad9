#!/bin/bash
# rocprofv3 evidence for the headline bench command: kernel trace + SQ / FETCH_SIZE / WRITE_SIZE passes (separate runs, as the MI355X
# guide prescribes), summarised into profiles/<tag>_rocprofv3_summary.md and <tag>_rocprofv3_summary_traffic.json.
# usage (on the GPU box): bash scripts/profile_round.sh r2 <git head>
TAG=${1:-r2}; HEAD=${2:-unknown}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python bench.py --steps 20 --warmup 3 --no-cpu-baseline"
O=gpurun_out/prof_$TAG
rocprofv3 --kernel-trace --stats -d $O/trace -o r -- $CMD > $O.trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY -d $O/pmc_sq -o r -- $CMD > $O.sq.log 2>&1
# (node-level products on the fp32-operand kernel in the traffic passes, so that every plane-GEMM dispatch is an edge-stage one)
MI_NODE_PLANES_MIN_ROWS=100000000 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o r -- $CMD > $O.fetch.log 2>&1
MI_NODE_PLANES_MIN_ROWS=100000000 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o r -- $CMD > $O.write.log 2>&1
python scripts/rocprof_summary.py gpurun_out/${TAG}_rocprofv3_summary.md $O/trace/r_results.db $O/pmc_sq/r_results.db $O/pmc_fetch/r_results.db $O/pmc_write/r_results.db > /dev/null
python - <<PY
import json
p = "gpurun_out/${TAG}_rocprofv3_summary_traffic.json"
d = json.load(open(p)); d["head"] = "$HEAD"; d["command"] = "$CMD"
json.dump(d, open(p, "w"), indent=1)
PY
rm -rf $O   # the raw databases are tens of MiB; the summaries are what is kept
python bench.py --steps 20 --warmup 3 > gpurun_out/${TAG}_bench_default_steps20.json 2>/dev/null
python bench.py > gpurun_out/${TAG}_bench_default.json 2>/dev/null
python bench.py --mode ft > gpurun_out/${TAG}_bench_finetune.json 2>/dev/null
tail -c 300 gpurun_out/${TAG}_bench_default.json; sed -n 7,16p gpurun_out/${TAG}_rocprofv3_summary.md | cut -c1-150
