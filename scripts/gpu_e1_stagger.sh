#!/bin/bash
# edge_gemm1b with the workgroups of the CUs' second slots starting late (MI_E1_STAGGER cycles): do out-of-phase co-resident workgroups overlap loop and epilogue?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in default st15 st30 st45; do
  if [ $v = default ]; then unset MI_LIB_PATH; else export MI_LIB_PATH=$GRAFT_REPO_ROOT/matinvent_amd/lib/variants/libmatinvent_hip_$v.so; fi
  echo "== $v"; python scripts/chains_timeline.py --steps 6 --warmup 2 --streams 1 2>&1 | grep -E "edge_gemm1b|edge_gemm2b" | cut -c1-130
  for st in 1 4; do echo -n "streams=$st: "; python bench.py --steps 20 --warmup 3 --streams $st --no-counters --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],3))"; done
done
