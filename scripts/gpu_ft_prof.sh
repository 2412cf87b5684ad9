#!/bin/bash
# kernel trace + PMC traffic passes of the pinned path's fine-tune line (BASELINE configs[2]) -> gpurun_out/<tag>_rocprofv3_summary_finetune.md (+ the per-kernel
# HBM-byte table <tag>_ft_traffic.md).  usage (through gpurun): bash scripts/gpu_ft_prof.sh [tag]
TAG=${1:-r6}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python bench.py --mode ft --steps 10 --warmup 3 --no-cpu-baseline --no-counters"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_ft -o ft -- $CMD > gpurun_out/prof_ft.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/prof_ft_fetch -o ft -- $CMD > gpurun_out/prof_ft_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/prof_ft_write -o ft -- $CMD > gpurun_out/prof_ft_write.log 2>&1
MI_SUMMARY_ROWS=44 python scripts/rocprof_summary.py gpurun_out/${TAG}_rocprofv3_summary_finetune.md gpurun_out/prof_ft/ft_results.db gpurun_out/prof_ft_fetch/ft_results.db gpurun_out/prof_ft_write/ft_results.db > /dev/null
rm -rf gpurun_out/prof_ft gpurun_out/prof_ft_fetch gpurun_out/prof_ft_write
python scripts/ft_traffic_table.py 4 > gpurun_out/${TAG}_ft_traffic.md 2> gpurun_out/${TAG}_ft_traffic.err
sed -n 1,50p gpurun_out/${TAG}_rocprofv3_summary_finetune.md | cut -c1-150
sed -n 1,30p gpurun_out/${TAG}_ft_traffic.md | cut -c1-160
