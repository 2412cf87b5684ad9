#!/bin/bash
# the pair-mode first edge GEMM: default (9: register-tile form b for large launches, k-loop under manual control) against the plane GEMM (0) and form E (4); alternating on one box
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_gpu_forward.py tests/test_gpu_sampler.py tests/test_gpu_saturation.py tests/test_gpu_train.py tests/test_gpu_pipeline.py -x -q 2>&1 | tail -40
for rep in 1 2; do for f in 9 0; do for st in 4 1; do MI_EDGE1_FUSED=$f timeout 600 python bench.py --steps 40 --warmup 5 --streams $st --no-cpu-baseline --no-counters 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('edge1=$f chains=$st', round(d['value'],3), 'structures/s', round(d['ms_per_step'],3), 'ms/step', 'frac', round(d['roofline']['frac'],3))"; done
MI_EDGE1_FUSED=$f python bench.py --mode ft --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('edge1=$f ft', round(d['value'],1))"
MI_EDGE1_FUSED=$f python bench.py --mode sample-default --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('edge1=$f sample-default', round(d['value'],2))"
done; done
