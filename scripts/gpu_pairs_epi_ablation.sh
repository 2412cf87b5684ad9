#!/bin/bash
# edge_gemm1b's pair epilogue, by ablation (timing only, wrong results): MI_DBG_PAIRS_SKIP bits 1 = one plane store per tile instead of all, 2 = no row gathers, 4 = no SiLU
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export MI_NODE_COLS=0
for rep in 1 2; do for v in default skip2 skip1 skip3 skip7; do
  if [ $v = default ]; then unset MI_LIB_PATH; else export MI_LIB_PATH=$GRAFT_REPO_ROOT/matinvent_amd/lib/variants/libmatinvent_hip_$v.so; fi
  for st in 4 1; do echo -n "$v streams=$st: "; timeout 300 python bench.py --steps 20 --warmup 3 --streams $st --no-counters --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],3))"; done
done; done
for v in default skip2 skip3; do
  if [ $v = default ]; then unset MI_LIB_PATH; else export MI_LIB_PATH=$GRAFT_REPO_ROOT/matinvent_amd/lib/variants/libmatinvent_hip_$v.so; fi
  echo "== $v phases (256 crystals)"; python scripts/edge2_phases.py 256 2>&1 | tail -5
done
