#!/bin/bash
# fine-tune line + its parity tests at HEAD; A/B of edge_mlp.2's weight gradient from plane sets (default) against fp32 rows (MI_TN128=259)
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_train.py -x -q 2>&1 | tail -4
for v in 3 259 3 259; do MI_TN128=$v python bench.py --mode ft --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ft tn128=$v', round(d['value'],1), round(d['ms_per_step'],2), 'ms/step')"; done
for v in 3 259 3 259; do MI_TN128=$v python bench.py --mode ft --steps 40 --warmup 5 --ft-groups 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ft groups=1 tn128=$v', round(d['value'],1), round(d['ms_per_step'],2), 'ms/step')"; done
