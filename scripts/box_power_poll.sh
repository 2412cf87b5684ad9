#!/bin/bash
# Power and clocks of the part WHILE the headline runs, through rocm-smi's metrics table (the hwmon files read stale values under this load: scripts/box_probe.sh).
# usage (through gpurun): bash scripts/box_power_poll.sh [bench args]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_box_power_poll.log
: > $O
python bench.py --steps 1000 --warmup 5 --no-cpu-baseline --no-counters "$@" > gpurun_out/r5_box_power_bench.json 2>/dev/null &
BP=$!
sleep 3   # (library load + set-up)
for i in $(seq 1 30); do
  { date +%s.%N; rocm-smi --showpower --showclocks --showtemp --showuse 2>/dev/null | grep -iE "power|sclk|mclk|fclk|junction|busy|use"; } >> $O 2>&1
  kill -0 $BP 2>/dev/null || break
  sleep 0.15
done
wait $BP
python -c "
import json; d=json.loads(open('gpurun_out/r5_box_power_bench.json').read().strip().splitlines()[-1]); print('bench:', round(d['value'],2), 'structures/s', round(d['ms_per_step'],3), 'ms/step')" >> $O
cat $O | cut -c1-150
