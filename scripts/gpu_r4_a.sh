#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | tail -25 > gpurun_out/r4a_pytest.log; tail -22 gpurun_out/r4a_pytest.log
for ch in 1 4 1 4; do timeout 600 python bench.py --mode mg-sample --steps 10 --warmup 2 --mg-chains $ch --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('mg chains=$ch', round(d['value'],4), 'structures/s', round(d['ms_per_step'],2), 'ms/step')"; done
bash scripts/gpu_m1_probe.sh 2>&1 | tee gpurun_out/r4a_m1_probe.log
