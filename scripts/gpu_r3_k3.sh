#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for w in 0 16 8 0 16 8; do echo -n "window=$w: "; MI_WGRAD_WINDOW=$w python bench.py --mode ft --no-cpu-baseline 2>/dev/null | tail -1 | grep -o '"value": [0-9.]*, "unit": "crystal-timesteps[^,]*'; done
MI_WGRAD_WINDOW=16 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_ft -o ft -- python bench.py --mode ft --steps 16 --warmup 16 --no-cpu-baseline > gpurun_out/prof_ft.log 2>&1
MI_SUMMARY_ROWS=24 python scripts/rocprof_summary.py gpurun_out/r3_rocprofv3_summary_finetune_w16.md gpurun_out/prof_ft/ft_results.db > /dev/null; rm -rf gpurun_out/prof_ft
sed -n 7,32p gpurun_out/r3_rocprofv3_summary_finetune_w16.md | cut -c1-150
