#!/bin/bash
# durations of the plane-set weight-gradient launches by grid size (dW2: 8 tiles x splits, dWff: 6 tiles x splits), one group
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for tiles in 512 256 1024; do
rm -rf /tmp/ftp2; MI_TN_TILES=$((tiles*3/2)) rocprofv3 --kernel-trace -d /tmp/ftp2 -o t --output-format csv -- python bench.py --mode ft --steps 10 --warmup 3 --ft-groups ${1:-1} --no-cpu-baseline > /dev/null 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob('/tmp/ftp2/**/*kernel_trace.csv', recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    if 'gemm_tn_planes' in n or 'tn_reduce' in n:
        d[(n.split('(')[0][-28:], r.get('Grid_Size', r.get('Grid_Size_X', '?')))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
print('target workgroups $tiles')
for k, v in sorted(d.items()):
    print('   %-30s grid %-8s %4d calls  avg %7.1f us' % (k[0], k[1], len(v), sum(v) / len(v)))
PY
done
