#!/bin/bash
# kernel traces of the headline command with the node chain as one launch / seven launches, four chains and one
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for cfg in "1 4" "0 4" "1 1" "0 1"; do
  set -- $cfg; nf=$1; st=$2
  O=gpurun_out/tr_nf${nf}_s${st}
  MI_NODE_FUSED=$nf rocprofv3 --kernel-trace --stats -d $O -o r -- python bench.py --steps 12 --warmup 3 --no-cpu-baseline --streams $st > $O.log 2>&1
  MI_SUMMARY_ROWS=14 python scripts/rocprof_summary.py gpurun_out/r3_trace_nf${nf}_s${st}.md $O/r_results.db > /dev/null
  grep -o '"value": [0-9.]*' $O.log | head -1
  sed -n 7,22p gpurun_out/r3_trace_nf${nf}_s${st}.md | cut -c1-170
  rm -rf $O
done
