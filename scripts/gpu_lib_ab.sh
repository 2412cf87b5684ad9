#!/bin/bash
# A/B of library variants (scripts/build_variant.py) against the shipped library on one box, alternating: headline over 100 steps (four chains) and one chain.
# usage: bash scripts/gpu_lib_ab.sh <tag> <variant> [<variant> ...]      ("now" = the shipped library)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; shift
O=gpurun_out/r5_${TAG}_ab.log
: > $O
one() { python bench.py "$@" --no-counters --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],2), round(d.get('ms_per_step',0),3))"; }
for rep in 1 2 3; do for v in now "$@"; do
  if [ $v = now ]; then unset MI_LIB_PATH; else export MI_LIB_PATH=$GRAFT_REPO_ROOT/matinvent_amd/lib/variants/libmatinvent_hip_$v.so; fi
  echo "$v streams=4 (100 steps): $(one --steps 100 --warmup 5)" >> $O
  echo "$v streams=1 (40 steps): $(one --steps 40 --warmup 3 --streams 1)" >> $O
done; done
cat $O
