cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/sdp; rocprofv3 --kernel-trace --stats -d /tmp/sdp -o t --output-format csv -- python bench.py --mode sample-default --steps 30 --no-cpu-baseline > /dev/null 2>&1
python - <<PY
import csv, glob
f = glob.glob('/tmp/sdp/**/*kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
for r in rows[:12]:
    print('   %-90s %6s calls  avg %8.1f us  %5.2f %%' % (r['Name'].split('(')[0][:90], r['Calls'], float(r['AverageNs']) / 1e3, float(r['Percentage'])))
PY
