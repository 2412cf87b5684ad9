#!/bin/bash
# the lean epilogue of gemm_rt (MatterGen-shaped dense layers): parity, which launches take it, and the sampler line with it on (1) / off (0), alternating on one box
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_gpu_mattergen.py tests/test_gpu_gemm.py "tests/test_gpu_forward.py::test_node_chain_launch_vs_the_seven_launch_form" -x -q 2>&1 | tail -4
MI_RT_TRACE=1 python scripts/rt_phases.py 0 8 2>&1 >/dev/null | grep "^gemm_rt" | sed 's/M=[0-9]* //' | sort | uniq -c | sort -rn | head -12
for rep in 1 2; do for l in 1 0; do for ch in 4 1; do MI_RT_LEAN=$l timeout 900 python bench.py --mode mg-sample --steps 6 --warmup 2 --mg-chains $ch --no-cpu-baseline --no-counters 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('mg lean=$l chains=$ch', round(d['value'],4), 'structures/s', round(d['ms_per_step'],2), 'ms/step')"; done; done; done
