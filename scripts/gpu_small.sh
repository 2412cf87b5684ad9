#!/bin/bash
# launch-bound regimes: a 4-crystal denoising step and the reference's default ragged B = 192 sampling batch, with kernel traces
cd $GRAFT_REPO_ROOT
python scripts/small_batch_step.py 10,7,4,10 200
python bench.py --mode sample-default --steps 20 --warmup 3 2>/dev/null | cut -c1-260
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_small -o s -- python scripts/small_batch_step.py 10,7,4,10 20 > gpurun_out/prof_small.log 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_sd -o s -- python bench.py --mode sample-default --steps 10 --warmup 2 > gpurun_out/prof_sd.log 2>&1
python - <<'PY'
import sqlite3
for db, steps in (("gpurun_out/prof_small/s_results.db", 30.0), ("gpurun_out/prof_sd/s_results.db", 12.0)):
    cur = sqlite3.connect(db).cursor()
    rows = [r for r in cur.execute("select name,total_calls,total_duration,average from top_kernels") if "spin_kernel" not in r[0]]
    tot = sum(r[1] for r in rows)
    print(db, "launches total", tot, "per step ~", tot / steps, "kernel time per step us", sum(r[2] for r in rows) / steps)
    for n, c, t, a in rows[:22]:
        print(f"{c:6d} {a:8.2f} us {t / steps:9.1f} us/step  {n[:90]}")
PY
rm -rf gpurun_out/prof_small gpurun_out/prof_sd
