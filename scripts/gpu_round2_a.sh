#!/bin/bash
# first GPU pass of round 2: full -m gpu suite (with tolerance report), default bench line, ft bench line
export MI_TOL_REPORT=1
python -m pytest tests -m gpu -x -q -s 2>&1 | tee gpurun_out/r2a_pytest.log | grep -v "^TOL" | tail -30
grep "^TOL" gpurun_out/r2a_pytest.log > gpurun_out/r2a_tol.log
unset MI_TOL_REPORT
python bench.py --steps 20 --warmup 3 > gpurun_out/r2a_bench20.json 2> gpurun_out/r2a_bench20.err; tail -c 600 gpurun_out/r2a_bench20.json
python bench.py --mode ft --steps 20 --warmup 2 > gpurun_out/r2a_bench_ft.json 2> gpurun_out/r2a_bench_ft.err; tail -c 600 gpurun_out/r2a_bench_ft.json
