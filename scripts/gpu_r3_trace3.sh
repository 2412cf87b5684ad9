#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for cfg in "1 1" "0 1" "1 4"; do
  set -- $cfg; e1=$1; st=$2
  O=gpurun_out/tr3_e${e1}_s${st}
  MI_EDGE1_FUSED=$e1 rocprofv3 --kernel-trace --stats -d $O -o r -- python bench.py --steps 12 --warmup 3 --no-cpu-baseline --streams $st > $O.log 2>&1
  MI_SUMMARY_ROWS=6 python scripts/rocprof_summary.py gpurun_out/r3_trace3_e${e1}_s${st}.md $O/r_results.db > /dev/null
  grep -o '"value": [0-9.]*' $O.log | head -1
  sed -n 7,14p gpurun_out/r3_trace3_e${e1}_s${st}.md | cut -c1-150
  rm -rf $O
done
