#!/bin/bash
# MatterGen-shaped fine-tune line (64 crystals: one chunk) + its kernel trace
cd $GRAFT_REPO_ROOT
python bench.py --mode mg-ft --mg-batch 64 2>&1 | tail -1 | cut -c1-300
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_mgft -o mg -- python bench.py --mode mg-ft --mg-batch 64 --steps 2 > gpurun_out/prof_mgft.log 2>&1
python scripts/rocprof_summary.py gpurun_out/r2_rocprofv3_summary_mattergen_finetune.md gpurun_out/prof_mgft/mg_results.db >/dev/null; rm -rf gpurun_out/prof_mgft
sed -n 7,26p gpurun_out/r2_rocprofv3_summary_mattergen_finetune.md | cut -c1-170
