#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -rf gpurun_out/ft1
rocprofv3 --kernel-trace --stats -d gpurun_out/ft1 -o ft -- python bench.py --mode ft --steps 12 --warmup 3 --ft-groups 1 --no-cpu-baseline > gpurun_out/ft1.log 2>&1
python - <<'PY'
import sqlite3, glob
db = glob.glob("gpurun_out/ft1/**/*_results.db", recursive=True)[0]
cur = sqlite3.connect(db).cursor()
rows = [r for r in cur.execute("select name,total_calls,total_duration,average from top_kernels") if "spin_kernel" not in r[0]]
tot = sum(r[2] for r in rows)
steps = 15.0
print(f"total kernel time {tot/1e3:.1f} ms over {steps:.0f} micro-steps = {tot/1e3/steps:.2f} ms per micro-step; {sum(r[1] for r in rows)/steps:.0f} dispatches per micro-step")
for n, c, t, a in rows[:45]:
    print(f"{c/steps:7.1f}/step {a:8.1f} us {t/1e3/steps:7.3f} ms/step {100*t/tot:5.1f} %  {n[:120]}")
PY
rm -rf gpurun_out/ft1
