#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for g in 1 2 4; do
  rm -rf gpurun_out/ft_tl
  rocprofv3 --kernel-trace -d gpurun_out/ft_tl -o ft -- python bench.py --mode ft --steps 12 --warmup 3 --ft-groups $g --no-cpu-baseline > gpurun_out/ft_tl.log 2>&1
  echo "groups=$g: $(grep -o '"value": [0-9.]*' gpurun_out/ft_tl.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/ft_tl.log | head -1)"
  python scripts/ft_timeline.py $(find gpurun_out/ft_tl -name "*_results.db" | head -1) 150 2>&1 | tee gpurun_out/r4_ft_timeline_groups$g.log
done
rm -rf gpurun_out/ft_tl
for g in 1 2 4; do python bench.py --mode ft --steps 20 --warmup 3 --ft-groups $g --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('unprofiled groups=$g', round(d['value'],1), round(d['ms_per_step'],2), 'ms/step')"; done
