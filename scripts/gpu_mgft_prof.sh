#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -rf gpurun_out/mgft
rocprofv3 --kernel-trace --stats -d gpurun_out/mgft -o ft -- python bench.py --mode mg-ft --steps 4 --no-cpu-baseline > gpurun_out/mgft.log 2>&1
grep -o '"value": [0-9.]*' gpurun_out/mgft.log | head -1
python - <<'PY'
import sqlite3, glob
db = glob.glob("gpurun_out/mgft/**/*_results.db", recursive=True)[0]
cur = sqlite3.connect(db).cursor()
rows = [r for r in cur.execute("select name,total_calls,total_duration,average from top_kernels") if "spin_kernel" not in r[0]]
tot = sum(r[2] for r in rows)
print(f"total kernel time {tot/1e3:.1f} ms, {sum(r[1] for r in rows)} dispatches")
for n, c, t, a in rows[:32]:
    print(f"{c:6d} {a:9.1f} us {t/1e3:9.2f} ms {100*t/tot:5.1f} %  {n[:110]}")
PY
rm -rf gpurun_out/mgft
