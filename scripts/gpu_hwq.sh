#!/bin/bash
# HIP maps streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues: do the concurrent groups / chains (+ their side streams) lose by sharing queues?
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for q in default 8 16; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  for g in 4 6; do python bench.py --mode ft --steps 50 --warmup 5 --ft-groups $g --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('hwq=$q ft groups=$g', round(d['value'],1), round(d['ms_per_step'],2), 'ms/step')"; done
  for st in 4 6; do python bench.py --steps 40 --warmup 5 --streams $st --no-cpu-baseline --no-counters 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('hwq=$q sampler chains=$st', round(d['value'],2), round(d['ms_per_step'],3), 'ms/step')"; done
done; done
