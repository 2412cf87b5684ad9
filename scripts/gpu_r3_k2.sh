#!/bin/bash
cd $GRAFT_REPO_ROOT
MI_WGRAD_WINDOW=0 python bench.py --mode ft --no-cpu-baseline 2>&1 | tail -12
timeout 1500 python -m pytest tests/test_gpu_train.py -x -q 2>&1 | grep -v "^  File" | head -30
