#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for cfg in "1 4" "1 1"; do
  set -- $cfg; e2=$1; st=$2
  O=gpurun_out/tr2_e${e2}_s${st}
  MI_EDGE2_FUSED=$e2 rocprofv3 --kernel-trace --stats -d $O -o r -- python bench.py --steps 12 --warmup 3 --no-cpu-baseline --streams $st > $O.log 2>&1
  MI_SUMMARY_ROWS=8 python scripts/rocprof_summary.py gpurun_out/r3_trace2_e${e2}_s${st}.md $O/r_results.db > /dev/null
  grep -o '"value": [0-9.]*' $O.log | head -1
  sed -n 7,16p gpurun_out/r3_trace2_e${e2}_s${st}.md | cut -c1-150
  rm -rf $O
done
timeout 900 python -m pytest tests/test_gpu_saturation.py -x -q 2>&1 | tail -3
timeout 1800 python -m pytest tests/test_gpu_mattergen.py -q -s -k "benchmark_size_four" 2>&1 | grep -E "MEASURED|passed|failed|Error|assert" | head -20
