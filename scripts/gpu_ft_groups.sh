#!/bin/bash
# fine-tune line against the number of concurrent crystal groups, alternating on one box
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for g in 4 2 3 5 6 8; do python bench.py --mode ft --steps 50 --warmup 5 --ft-groups $g --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ft groups=$g', round(d['value'],1), round(d['ms_per_step'],2), 'ms/step')"; done; done
