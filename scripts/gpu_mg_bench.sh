#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python bench.py --mode mg-sample --steps 4 --warmup 1 --mg-batch 64 2>&1 | tail -3
python bench.py --mode mg-sample --steps 4 --warmup 1 2>&1 | tail -3
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_mg -o mg -- python bench.py --mode mg-sample --steps 2 --warmup 1 > gpurun_out/prof_mg.log 2>&1
ls gpurun_out/prof_mg | head
