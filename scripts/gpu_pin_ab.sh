#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
V=${1:-pin}
timeout 900 env MI_LIB_PATH=$GRAFT_REPO_ROOT/matinvent_amd/lib/variants/libmatinvent_hip_$V.so python -m pytest tests/test_gpu_forward.py -x -q -m gpu -k "node_chain or full_size or north_star" 2>&1 | tail -2
for rep in 1 2 3; do for v in now $V; do
  if [ $v = now ]; then unset MI_LIB_PATH; else export MI_LIB_PATH=$GRAFT_REPO_ROOT/matinvent_amd/lib/variants/libmatinvent_hip_$v.so; fi
  for st in 4 1; do echo -n "$v streams=$st: "; python bench.py --steps 20 --warmup 3 --streams $st --no-counters --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],3))"; done
done; done
export MI_LIB_PATH=$GRAFT_REPO_ROOT/matinvent_amd/lib/variants/libmatinvent_hip_$V.so; python scripts/edge2_phases.py 256 2>&1 | tail -14
