#!/bin/bash
# the reference's default fine-tune set (18 crystals, stacked timesteps): which of this round's training-path changes it likes
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for cfg in "base:" "node_train0:MI_NODE_TRAIN=0" "tn259:MI_TN128=259" "both:MI_NODE_TRAIN=0 MI_TN128=259"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs python bench.py --mode ft-default --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['value'],1), d['unit'], round(d['ms_per_step'],4), 'ms/step')"
done; done
