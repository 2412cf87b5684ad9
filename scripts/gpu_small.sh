#!/bin/bash
cd $GRAFT_REPO_ROOT
python scripts/small_batch_step.py 10,7,4,10 200
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_small -o s -- python scripts/small_batch_step.py 10,7,4,10 20 > gpurun_out/prof_small.log 2>&1
python - <<'PY'
import sqlite3
cur = sqlite3.connect("gpurun_out/prof_small/s_results.db").cursor()
rows = list(cur.execute("select name,total_calls,total_duration,average from top_kernels"))
tot = sum(r[1] for r in rows)
print("launches total", tot, "per step ~", tot / 30.0)
for n, c, t, a in rows[:40]:
    print(f"{c:6d} {a:8.2f} us  {n[:100]}")
PY
