#!/bin/bash
# register-tile kernels: every memory operation of the k-loops under manual control (MI_ASM_LDS=2, default) against inline-asm LDS reads only (1) and compiler-visible reads + __syncthreads() (0); builds alternate on one box
cd $GRAFT_REPO_ROOT
python -m matinvent_amd.build --force 2>&1 | tail -1; timeout 900 python -m pytest tests/test_gpu_forward.py tests/test_gpu_gemm.py tests/test_gpu_sampler.py -x -q 2>&1 | tail -2; timeout 900 python -m pytest tests/test_gpu_mattergen.py -x -q -k "large_tile or benchmark_size_forward" 2>&1 | tail -2
run() {
  MI_EXTRA_FLAGS="$2" python -m matinvent_amd.build --force 2>&1 | grep -v "recognized feature\|^/opt/rocm" | tail -1
  for st in 4 1; do timeout 600 python bench.py --steps 40 --warmup 5 --streams $st --no-cpu-baseline --no-counters 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1 sampler chains=$st', round(d['value'],3), 'structures/s', round(d['ms_per_step'],3), 'ms/step', 'avg_launch_ms', round(d['roofline']['avg_launch_ms'],4))"; done
  timeout 900 python bench.py --mode mg-sample --steps 6 --warmup 2 --mg-chains 4 --no-cpu-baseline --no-counters 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1 mg chains=4', round(d['value'],4), 'structures/s')"
  timeout 900 python bench.py --mode ft --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1 ft', round(d['value'],1))"
}
run asm2 ""
run asm1 "-DMI_ASM_LDS=1"
run asm0 "-DMI_ASM_LDS=0"
run asm2b ""
run asm0b "-DMI_ASM_LDS=0"
