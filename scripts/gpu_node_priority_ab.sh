#!/bin/bash
# node-level kernels on a high-priority helper stream per chain: identical results, then the headline with it off / on
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_gemm.py -m gpu -x -q -k "network" 2>&1 | tail -3
for rep in 1 2; do for hi in 0 1; do echo "node priority $hi"; MI_NODE_PRIORITY=$hi python bench.py --steps 60 --warmup 3 --no-cpu-baseline 2>/dev/null | cut -c1-140; done; done
for hi in 0 1; do echo "node priority $hi, 1 chain"; MI_NODE_PRIORITY=$hi python bench.py --steps 40 --warmup 3 --streams 1 --no-cpu-baseline 2>/dev/null | cut -c1-140; done
