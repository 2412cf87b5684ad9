#!/bin/bash
# kernel trace of the pinned path's fine-tune line (BASELINE configs[2]) -> gpurun_out/r5_rocprofv3_summary_finetune.md
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_ft -o ft -- python bench.py --mode ft --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/prof_ft.log 2>&1
MI_SUMMARY_ROWS=40 python scripts/rocprof_summary.py gpurun_out/r5_rocprofv3_summary_finetune.md gpurun_out/prof_ft/ft_results.db > /dev/null; rm -rf gpurun_out/prof_ft
sed -n 1,60p gpurun_out/r5_rocprofv3_summary_finetune.md | cut -c1-175
