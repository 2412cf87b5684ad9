#!/bin/bash
# generic A/B: this tree's library against variants/libmatinvent_hip_$1.so -- parity files first, then headline / one chain / default batch alternating on one box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
V=${1:-before}
timeout 1500 python -m pytest tests/test_gpu_forward.py tests/test_gpu_sampler.py -x -q -m gpu 2>&1 | tail -2
for rep in 1 2 3; do for v in $V now; do
  if [ $v = now ]; then unset MI_LIB_PATH; else export MI_LIB_PATH=$GRAFT_REPO_ROOT/matinvent_amd/lib/variants/libmatinvent_hip_$v.so; fi
  for st in 4 1; do echo -n "$v streams=$st: "; python bench.py --steps 20 --warmup 3 --streams $st --no-counters --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],3))"; done
  echo -n "$v sample-default: "; python bench.py --mode sample-default --no-counters --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],2))"
done; done
for v in $V now; do
  if [ $v = now ]; then unset MI_LIB_PATH; else export MI_LIB_PATH=$GRAFT_REPO_ROOT/matinvent_amd/lib/variants/libmatinvent_hip_$v.so; fi
  echo "== $v"; python scripts/node_chain_phases.py 64 2>&1 | sed -n 2,8p
done
