#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_mattergen.py -q 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_mg3 -o mg -- python bench.py --mode mg-sample --steps 2 --warmup 1 > gpurun_out/prof_mg3.log 2>&1
python scripts/rocprof_summary.py /tmp/x.md gpurun_out/prof_mg3/mg_results.db >/dev/null; sed -n 7,14p /tmp/x.md | cut -c1-150
