#!/bin/bash
# fine-tune line against the number of workgroups a weight-gradient contraction is split into (MI_TN_TILES; partial tiles = traffic), alternating on one box
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_train.py -x -q 2>&1 | tail -3
for rep in 1 2; do for t in 768 384 256 128; do MI_TN_TILES=$t python bench.py --mode ft --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ft tiles=$t', round(d['value'],1), round(d['ms_per_step'],2), 'ms/step')"; done; done
for t in 768 384 256; do MI_TN_TILES=$t python bench.py --mode ft --steps 40 --warmup 5 --ft-groups 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ft groups=1 tiles=$t', round(d['value'],1), round(d['ms_per_step'],2), 'ms/step')"; done
