#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_gpu_mattergen.py -q -s -k "benchmark_size_four or benchmark_size_fine" 2>&1 | grep -v "^$" | tail -25
timeout 1800 python -m pytest tests/test_gpu_saturation.py tests/test_gpu_multirank.py tests/test_gpu_train.py tests/test_gpu_gemm.py -x -q 2>&1 | tail -12
for mode in sample-default ft-default mg-ft; do
  timeout 600 python bench.py --mode $mode > gpurun_out/r3_e_$mode.json 2> gpurun_out/r3_e_$mode.err; echo "$mode rc=$?"; tail -c 1500 gpurun_out/r3_e_$mode.json; tail -3 gpurun_out/r3_e_$mode.err
done
timeout 600 python bench.py --mode mg-sample --steps 4 > gpurun_out/r3_e_mg.json 2> gpurun_out/r3_e_mg.err; tail -c 2500 gpurun_out/r3_e_mg.json; tail -3 gpurun_out/r3_e_mg.err
