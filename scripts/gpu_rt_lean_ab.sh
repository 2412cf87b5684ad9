#!/bin/bash
# the lean epilogue of gemm_rt (MatterGen-shaped dense layers): parity, phase clock, and the sampler line with it persistent (2) / on (1) / off (0), alternating on one box
cd $GRAFT_REPO_ROOT
MI_RT_LEAN=2 timeout 2400 python -m pytest tests/test_gpu_mattergen.py -x -q -k "large_tile or benchmark_size_forward or benchmark_size_crystals or without_a_host" 2>&1 | tail -3
MI_RT_LEAN=2 python scripts/rt_phases.py 0 8 2>/dev/null | grep -v amdgpu.ids
MI_RT_LEAN=2 python scripts/rt_phases.py 1 8 2>/dev/null | grep -v amdgpu.ids
for rep in 1 2; do for l in 2 1 0; do for ch in 4 1; do MI_RT_LEAN=$l timeout 900 python bench.py --mode mg-sample --steps 6 --warmup 2 --mg-chains $ch --no-cpu-baseline --no-counters 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('mg lean=$l chains=$ch', round(d['value'],4), 'structures/s', round(d['ms_per_step'],2), 'ms/step')"; done; done; done
