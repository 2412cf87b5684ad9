#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_forward.py -x -q -k "node_chain or full_size_forward_vs" 2>&1 | tail -3
for v in 1 3; do python scripts/edge2_phases.py 64 $v 2>&1 | tail -5; python scripts/edge2_phases.py 256 $v 2>&1 | tail -5; done
for cfg in "0 4" "1 4" "3 4" "1 4" "3 4" "1 1" "3 1" "1 2" "1 3"; do
  set -- $cfg
  echo "edge2=$1 streams=$2: $(MI_EDGE2_FUSED=$1 timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --streams $2 2>/dev/null | cut -c75-110)"
done
