#!/bin/bash
cd $GRAFT_REPO_ROOT
python bench.py --mode mg-ft --mg-batch 64 2>&1 | tail -2 | cut -c1-600
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_mgft -o mg -- python bench.py --mode mg-ft --mg-batch 64 --steps 2 > gpurun_out/prof_mgft.log 2>&1
python scripts/rocprof_summary.py gpurun_out/prof_mgft.md gpurun_out/prof_mgft/mg_results.db >/dev/null; sed -n 7,24p gpurun_out/prof_mgft.md | cut -c1-160
