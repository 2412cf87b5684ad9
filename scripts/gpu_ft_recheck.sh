#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for v in before now; do
  if [ $v = now ]; then unset MI_LIB_PATH; else export MI_LIB_PATH=$GRAFT_REPO_ROOT/matinvent_amd/lib/variants/libmatinvent_hip_$v.so; fi
  echo -n "$v ft: "; python bench.py --mode ft --no-cpu-baseline --no-counters 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],3))"
done; done
for v in before now; do
  if [ $v = now ]; then unset MI_LIB_PATH; else export MI_LIB_PATH=$GRAFT_REPO_ROOT/matinvent_amd/lib/variants/libmatinvent_hip_$v.so; fi
  for st in 4 1; do echo -n "$v streams=$st: "; python bench.py --steps 20 --warmup 3 --streams $st --no-counters --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],3))"; done
done
unset MI_LIB_PATH; python scripts/edge2_phases.py 256 2>&1 | tail -14
