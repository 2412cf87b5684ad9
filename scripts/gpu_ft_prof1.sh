#!/bin/bash
# per-kernel averages of the fine-tune line on ONE group (kernels run one after the other: a kernel's average is its own time)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 3 259; do
rm -rf /tmp/ftp1; MI_TN128=$v rocprofv3 --kernel-trace --stats -d /tmp/ftp1 -o t --output-format csv -- python bench.py --mode ft --steps 10 --warmup 3 --ft-groups 1 --no-cpu-baseline > /dev/null 2>&1
python - <<PY
import csv, glob
f = glob.glob('/tmp/ftp1/**/*kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('MI_TN128=$v: kernel time per micro-step %.2f ms' % (tot / 1e6 / 13))
for r in rows[:16]:
    print('   %-78s %5s calls  avg %8.1f us  %5.2f %%' % (r['Name'].split('(')[0][:78], r['Calls'], float(r['AverageNs']) / 1e3, float(r['Percentage'])))
PY
done
