#!/bin/bash
# Where a short timed window loses against the steady state: the headline at several window lengths and warm-up lengths on one box.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_window_sweep.log
: > $O
for rep in 1 2; do
for cfg in "20 5" "20 50" "20 200" "40 5" "100 5" "200 5" "800 5"; do
  set -- $cfg
  python bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-counters $EXTRA 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('steps $1 warmup $2:', round(d['value'],2), 'structures/s', round(d['ms_per_step'],3), 'ms/step  window', round(d['ms_per_step']*d['steps'],2), 'ms  avg_launch', round(d['roofline']['avg_launch_ms'],4), 'busy share', round(d['roofline']['stage_busy_share_of_timed_region'],3))" >> $O
done
done
cat $O
