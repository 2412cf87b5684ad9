#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_gemm.py tests/test_gpu_forward.py tests/test_gpu_sampler.py -m gpu -x -q 2>&1 | tail -3
python scripts/small_batch_step.py 10,7,4,10 200
python scripts/default_batch_streams.py | tail -4
for rep in 1 2 3; do python bench.py --steps 60 --warmup 3 --no-cpu-baseline 2>/dev/null | cut -c1-140; done
python bench.py --mode ft --steps 60 --warmup 3 --no-cpu-baseline 2>/dev/null | cut -c1-140
