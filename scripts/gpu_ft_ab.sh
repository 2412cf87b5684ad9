#!/bin/bash
# fine-tune line + its parity tests at HEAD (same box A/B is done by stashing: here just HEAD numbers, three runs)
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_train.py -x -q 2>&1 | tail -4
for i in 1 2 3; do python bench.py --mode ft --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ft', round(d['value'],1), round(d['ms_per_step'],2), 'ms/step')"; done
for i in 1 2; do python bench.py --mode ft --steps 40 --warmup 5 --ft-groups 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ft groups=1', round(d['value'],1), round(d['ms_per_step'],2), 'ms/step')"; done
