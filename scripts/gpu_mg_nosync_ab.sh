#!/bin/bash
# MatterGen-shaped sampler with / without the host round trip per evaluation (same box), + its GPU tests
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_mattergen.py -x -q --durations=5 2>&1 | tail -15
for ns in 0 1 0 1; do
  MI_MG_NOSYNC=$ns timeout 600 python bench.py --mode mg-sample --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('nosync=$ns', round(d['value'],4), 'structures/s', round(d['ms_per_step'],2), 'ms/step', d['config']['edges_first_step'], d['config']['edges_last_step'], d['config']['final_state_finite'])"
done
