#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_mattergen.py -q 2>&1 | tail -3
python bench.py --mode mg-sample --steps 6 --warmup 1 2>/dev/null | tail -1 | cut -c1-330
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_mg3 -o mg -- python bench.py --mode mg-sample --steps 2 --warmup 1 > gpurun_out/prof_mg3.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/prof_mg3_f -o mg -- python bench.py --mode mg-sample --steps 2 --warmup 1 > gpurun_out/prof_mg3_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/prof_mg3_w -o mg -- python bench.py --mode mg-sample --steps 2 --warmup 1 > gpurun_out/prof_mg3_w.log 2>&1
ls gpurun_out/prof_mg3*
