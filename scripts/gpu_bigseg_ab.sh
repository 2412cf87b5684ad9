#!/bin/bash
# second edge GEMM on the 256 x 256 LDS-DMA kernel: bit-identity in the network, then the headline with it off / on (4 chains and 1 chain)
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_gemm.py -m gpu -x -q -k "network" 2>&1 | tail -3
for seg in 0 20000 0 20000; do echo "big-seg min rows $seg, 4 chains"; MI_PLANES_BIG_SEG=$seg python bench.py --steps 40 --warmup 3 --no-cpu-baseline 2>/dev/null | cut -c1-140; done
for seg in 0 20000; do echo "big-seg min rows $seg, 1 chain"; MI_PLANES_BIG_SEG=$seg python bench.py --steps 40 --warmup 3 --streams 1 --no-cpu-baseline 2>/dev/null | cut -c1-140; done
for seg in 0 20000; do echo "big-seg min rows $seg, 2 chains"; MI_PLANES_BIG_SEG=$seg python bench.py --steps 40 --warmup 3 --streams 2 --no-cpu-baseline 2>/dev/null | cut -c1-140; done
