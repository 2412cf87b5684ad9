#!/bin/bash
# deferred node-level weight gradients: parity tests + A/B of the fine-tune line
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_train.py -x -q 2>&1 | tail -8
for w in 0 16 0 16; do echo -n "window=$w: "; MI_WGRAD_WINDOW=$w python bench.py --mode ft --no-cpu-baseline 2>/dev/null | tail -1 | grep -o '"value": [0-9.]*, "unit": "crystal-timesteps[^,]*'; done
