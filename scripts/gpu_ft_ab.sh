#!/bin/bash
# fine-tune line + its parity tests at HEAD; A/B of the training forward's node-level form (MI_NODE_TRAIN=0: seven launches per layer)
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_train.py -x -q 2>&1 | tail -4
for nt in 1 0 1 0; do MI_NODE_TRAIN=$nt python bench.py --mode ft --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ft node_train=$nt', round(d['value'],1), round(d['ms_per_step'],2), 'ms/step')"; done
for nt in 1 0; do MI_NODE_TRAIN=$nt python bench.py --mode ft --steps 40 --warmup 5 --ft-groups 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ft groups=1 node_train=$nt', round(d['value'],1), round(d['ms_per_step'],2), 'ms/step')"; done
