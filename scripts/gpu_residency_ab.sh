#!/bin/bash
# room for the node-level launches beside the resident edge-GEMM workgroups: mode 1 (default) vs 3 (second GEMM on the LDS-DMA form, two
# per CU; small launches register-staged, 32 KiB) vs 4 (3 + small launches on the 128-register build)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for mode in 1 3 4; do echo "planes-dma mode $mode"; MI_PLANES_DMA=$mode python bench.py --steps 60 --warmup 3 --no-cpu-baseline 2>/dev/null | cut -c1-140; done; done
