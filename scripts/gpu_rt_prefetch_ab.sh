#!/bin/bash
# gemm_rt_kernel<EXT>: the epilogue's row-wise operands touched MI_RT_PF_AT k-tiles before the loop ends (0 = never); MatterGen-shaped sampler, same box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() {
  MI_EXTRA_FLAGS="-DMI_RT_PF_AT=$1" python -m matinvent_amd.build --force 2>&1 | grep -v "recognized feature\|^/opt/rocm" | tail -1 > /dev/null
  for ch in 1 4; do for i in 1 2; do timeout 600 python bench.py --mode mg-sample --steps 10 --warmup 2 --mg-chains $ch --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('pf_at=$1 chains=$ch', round(d['value'],4), 'structures/s', round(d['ms_per_step'],2), 'ms/step')"; done; done
}
run 0
run 3
run 5
run 8
run 0
python -m matinvent_amd.build --force > /dev/null 2>&1
timeout 1200 python -m pytest tests/test_gpu_mattergen.py -x -q -k "benchmark_size and not fine_tune" 2>&1 | tail -3
bash scripts/gpu_chains_queues.sh
