#!/bin/bash
# whole GPU suite + the driver-shaped headline run + the MatterGen-shaped sampler line at HEAD
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/full_pytest.log
python bench.py --steps 20 --warmup 3 > gpurun_out/full_bench20.json 2> gpurun_out/full_bench20.err
python bench.py --mode mg-sample --steps 6 --warmup 1 > gpurun_out/full_mg.json 2> gpurun_out/full_mg.err
tail -3 gpurun_out/full_pytest.log
cut -c1-200 gpurun_out/full_bench20.json
cut -c1-200 gpurun_out/full_mg.json
