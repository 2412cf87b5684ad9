#!/bin/bash
# MatterGen-shaped path: its GPU tests, then the sampler line (same steps / warmup as the whole-suite check: the edge count drifts along
# the chain), with the lean inference switches ablated, and the headline line (the plane-set kernel is shared with the pinned path)
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_mattergen.py -x -q 2>&1 | tail -4
for mask in 15 0 1 9 11 13; do
  echo "lean mask $mask"
  MI_MG_LEAN=$mask python bench.py --mode mg-sample --steps 6 --warmup 1 --no-cpu-baseline 2>gpurun_out/mg_ab.err | cut -c1-240
done
python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | cut -c1-200
