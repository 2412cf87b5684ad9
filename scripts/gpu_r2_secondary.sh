#!/bin/bash
cd $GRAFT_REPO_ROOT
for mode in mg-sample sample-default ft-default; do
  python bench.py --mode $mode 2> gpurun_out/r2_bench_$mode.err | tail -1 > gpurun_out/r2_bench_$mode.json
  cut -c1-400 gpurun_out/r2_bench_$mode.json; tail -2 gpurun_out/r2_bench_$mode.err
done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_mg2 -o mg -- python bench.py --mode mg-sample --steps 3 --warmup 1 > gpurun_out/prof_mg2.log 2>&1
