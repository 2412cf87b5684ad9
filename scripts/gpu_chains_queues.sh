#!/bin/bash
# concurrent chains beyond four: does the cliff at 6 / 8 chains (41.8 / 35.1 structures/s) come from the runtime's four hardware queues?
cd $GRAFT_REPO_ROOT
for q in "" 8; do for st in 4 6 8; do
  GPU_MAX_HW_QUEUES=$q timeout 600 python bench.py --steps 40 --warmup 5 --streams $st --no-cpu-baseline --no-counters 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('GPU_MAX_HW_QUEUES=${q:-default} chains=$st', round(d['value'],3), 'structures/s', round(d['ms_per_step'],3), 'ms/step')"
done; done
