#!/bin/bash
# round 4: is the concurrent-chain nondeterminism of the MatterGen-shaped forwards the packed-fp32 fault of scripts/force_fwd_repro.hip?
# same box: the library as shipped vs built without v_pk_*_f32 instructions; 120 trials of four concurrent forwards each + the headline line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() {   # $1 = tag, $2 = extra flags
  MI_EXTRA_FLAGS="$2" python -m matinvent_amd.build --force 2>&1 | grep -v "recognized feature\|^/opt/rocm" | tail -2
  MI_CONC_TRIALS=120 timeout 900 python scripts/mg_concurrent_forward_check.py 2>&1 | grep -v Warning > gpurun_out/r4_conc_$1.log
  echo "$1: concurrent trials identical: $(grep -c 'concurrent: identical' gpurun_out/r4_conc_$1.log) of $(grep -c 'concurrent:' gpurun_out/r4_conc_$1.log)"
  grep 'concurrent:' gpurun_out/r4_conc_$1.log | grep -v identical | cut -c1-200 | head -8
  for i in 1 2; do timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-counters 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1 headline', round(d['value'],3), 'structures/s', round(d['ms_per_step'],3), 'ms/step')"; done
}
run default ""
run nopk "-Xclang -target-feature -Xclang -packed-fp32-ops"
run default2 ""
