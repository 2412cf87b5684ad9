#!/bin/bash
# A/B of ONE environment knob of bench.py (_apply_env_knobs) on one box, alternating: headline (4 chains, 100 steps), one chain, the reference's default batch.
# usage: bash scripts/gpu_env_ab.sh NAME A_VALUE B_VALUE [tag]     (an empty value = the library's default)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
N=$1; A=$2; B=$3; TAG=${4:-$1}
O=gpurun_out/r5_${TAG}_ab.log
: > $O
one() { python bench.py "$@" --no-counters --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],2), round(d.get('ms_per_step',0),3))"; }
for rep in 1 2 3; do for v in "$A" "$B"; do
  export $N="$v"
  echo "$N=$v streams=4 (100 steps): $(one --steps 100 --warmup 5)" >> $O
  echo "$N=$v streams=4 (20 steps): $(one --steps 20 --warmup 5)" >> $O
  echo "$N=$v streams=1: $(one --steps 20 --warmup 3 --streams 1)" >> $O
  echo "$N=$v sample-default: $(one --mode sample-default)" >> $O
done; done
cat $O
