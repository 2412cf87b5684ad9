#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_mg3 -o mg -- python bench.py --mode mg-sample --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof_mg3.log 2>&1
python scripts/rocprof_summary.py /tmp/x.md gpurun_out/prof_mg3/mg_results.db >/dev/null; sed -n 7,22p /tmp/x.md | cut -c1-170
rm -rf gpurun_out/prof_mg3
