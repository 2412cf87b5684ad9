#!/bin/bash
# A/B of the epilogue instruction diet on one box, alternating builds: new / r3 (MI_AB_SPLIT_R3 + MI_AB_PAIRS_R3: round 3's plane split and
# pair-epilogue addressing / arithmetic) / new; per build the headline on four chains and on one, and the per-kernel averages of a short trace
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
run() {
  MI_EXTRA_FLAGS="$2" python -m matinvent_amd.build --force 2>&1 | grep -v "recognized feature\|^/opt/rocm" | tail -1
  for st in 4 1; do for i in 1 2; do timeout 600 python bench.py --steps 40 --warmup 5 --streams $st --no-cpu-baseline --no-counters 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1 chains=$st', round(d['value'],3), 'structures/s', round(d['ms_per_step'],3), 'ms/step', 'avg_launch_ms', round(d['roofline']['avg_launch_ms'],4))"; done; done
  rm -rf /tmp/prof_$1; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$1 -o t -- python bench.py --steps 10 --warmup 3 --streams 1 --no-cpu-baseline --no-counters > /dev/null 2>&1
  python - <<PY
import csv, glob
f = glob.glob('/tmp/prof_$1/**/*kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
for r in rows[:6]:
    print('   $1 one-chain trace:', r['Name'][:70], r['Calls'], 'avg us', round(float(r['AverageNs'])/1e3, 1), r['Percentage'])
PY
}
run new ""
run r3 "-DMI_AB_SPLIT_R3 -DMI_AB_PAIRS_R3"
run new2 ""
run r3b "-DMI_AB_SPLIT_R3 -DMI_AB_PAIRS_R3"
