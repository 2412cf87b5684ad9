#!/bin/bash
# round 4, review item 1: what would an L2-resident M1 buy at most?  Timing builds with WRONG results (MI_DBG_PAIRS_SKIP): 16 = every M1 store
# lands in the first 1024 rows (no write-back), 32 = the second edge GEMM reads M1 from its first eight row tiles (L2 hits), 48 = both.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() {
  MI_EXTRA_FLAGS="$2" python -m matinvent_amd.build --force 2>&1 | grep -v "recognized feature\|^/opt/rocm" | tail -1
  for st in 4 1; do for i in 1 2; do timeout 600 python bench.py --steps 40 --warmup 5 --streams $st --no-cpu-baseline --no-counters 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1 chains=$st', round(d['value'],3), 'structures/s', round(d['ms_per_step'],3), 'ms/step')"; done; done
}
run base ""
run skip16 "-DMI_DBG_PAIRS_SKIP=16"
run skip32 "-DMI_DBG_PAIRS_SKIP=32"
run skip48 "-DMI_DBG_PAIRS_SKIP=48"
run base2 ""
