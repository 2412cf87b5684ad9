#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_forward.py -x -q -k "node_chain_launch" 2>&1 | tail -4
for rep in 1 2; do for m in 0 1 2; do echo -n "edge1_fused=$m streams=4: "; MI_EDGE1_FUSED=$m python bench.py --steps 60 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | grep -o '"value": [0-9.]*, "unit": "struct[^,]*'; done; done
for m in 0 2; do echo -n "edge1_fused=$m streams=1: "; MI_EDGE1_FUSED=$m python bench.py --steps 60 --warmup 3 --streams 1 --no-cpu-baseline 2>/dev/null | tail -1 | grep -o '"value": [0-9.]*, "unit": "struct[^,]*'; done
