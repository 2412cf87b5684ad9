#!/bin/bash
# the pair epilogue's index rows and the node chain's row pointers as ONE batch of loads (this tree) against the library before it
# (variants/libmatinvent_hip_before.so): parity files, then alternating on one box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_forward.py tests/test_gpu_sampler.py tests/test_gpu_train.py -x -q -m gpu 2>&1 | tail -3
for rep in 1 2 3; do for v in before now; do
  if [ $v = now ]; then unset MI_LIB_PATH; else export MI_LIB_PATH=$GRAFT_REPO_ROOT/matinvent_amd/lib/variants/libmatinvent_hip_$v.so; fi
  for st in 4 1; do echo -n "$v streams=$st: "; python bench.py --steps 20 --warmup 3 --streams $st --no-counters --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],3))"; done
  echo -n "$v sample-default: "; python bench.py --mode sample-default --no-counters --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],2))"
done; done
for v in before now; do
  if [ $v = now ]; then unset MI_LIB_PATH; else export MI_LIB_PATH=$GRAFT_REPO_ROOT/matinvent_amd/lib/variants/libmatinvent_hip_$v.so; fi
  echo "== $v"; python scripts/edge2_phases.py 256 2>&1 | tail -5; MI_NODE_COLS=0 MI_NODE_SPLIT=0 python scripts/node_chain_phases.py 64 2>&1 | sed -n 2,8p
  echo -n "$v ft: "; python bench.py --mode ft --no-cpu-baseline --no-counters 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],3))"
done
