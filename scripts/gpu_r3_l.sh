#!/bin/bash
# register-tile form of the large plane products: bit-identity + MatterGen-shaped suite, A/B on the mg-sample line; ft roofline by group count
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_gemm.py -x -q 2>&1 | tail -4
timeout 1500 python -m pytest tests/test_gpu_mattergen.py -x -q -k "not fine_tune_window" 2>&1 | tail -4
for m in 0 1 2 0 1 2; do echo -n "planes_rt=$m: "; MI_PLANES_RT=$m python bench.py --mode mg-sample --no-cpu-baseline 2>/dev/null | tail -1 | grep -o '"value": [0-9.]*, "unit": "[^"]*"'; done
for g in 3 4; do echo -n "ft groups=$g: "; python bench.py --mode ft --ft-groups $g --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['roofline']['stage_busy_ms'], d['roofline']['launches'])"; done
echo -n "ft 20 steps: "; python bench.py --mode ft --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['roofline']['stage_busy_ms'], d['roofline']['launches'])"
