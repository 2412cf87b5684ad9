#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_forward.py tests/test_gpu_sampler.py -x -q 2>&1 | tail -3
python scripts/node_chain_phases.py 64 2>&1 | tail -12
for cfg in 4 4 1 1; do
  echo "streams=$cfg: $(timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --streams $cfg 2>/dev/null | cut -c75-110)"
done
echo "sample-default: $(timeout 300 python bench.py --mode sample-default --steps 100 --no-cpu-baseline 2>/dev/null | cut -c100-150)"
echo "ft: $(timeout 300 python bench.py --mode ft --no-cpu-baseline 2>/dev/null | cut -c55-100)"
