#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_forward.py -x -q -k "node_chain or full_size_forward_vs or north_star" 2>&1 | tail -8
for cfg in "0 4" "1 4" "0 4" "1 4" "0 1" "1 1"; do
  set -- $cfg
  echo "edge1=$1 streams=$2: $(MI_EDGE1_FUSED=$1 timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --streams $2 2>/dev/null | cut -c75-110)"
done
for e1 in 0 1; do echo "ft edge1=$e1: $(MI_EDGE1_FUSED=$e1 timeout 300 python bench.py --mode ft --no-cpu-baseline 2>/dev/null | cut -c55-100)"; done
