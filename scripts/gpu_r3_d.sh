#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
free -g | head -2; nproc
timeout 2400 python -m pytest tests/test_gpu_mattergen.py tests/test_gpu_saturation.py tests/test_gpu_multirank.py tests/test_gpu_train.py -x -q --durations=12 2>&1 | tail -40 > gpurun_out/r3_d_pytest.log
tail -30 gpurun_out/r3_d_pytest.log
python scripts/node_chain_phases.py 64 2>&1 | tail -12
