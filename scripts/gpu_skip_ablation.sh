#!/bin/bash
# What a chain's serial path and the chip's occupancy cost each other: the headline with parts of the forward SKIPPED (timing only, results garbage).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { "$@" timeout 300 python bench.py --steps 20 --warmup 3 --no-counters --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
l=sys.stdin.read()
try:
    d=json.loads(l); print(round(d['value'],2), round(d['ms_per_step'],3))
except Exception as e: print('failed:', l[-300:])"; }
for sk in 0 1 2 4 6 7; do echo "== MI_SKIP=$sk (cols 0), 4 streams"; MI_NODE_COLS=0 MI_SKIP=$sk run env; done
for st in 1 2 3; do for m in 0 1 2; do echo "== streams $st MI_NODE_COLS=$m"; MI_NODE_COLS=$m timeout 300 python bench.py --steps 20 --warmup 3 --no-counters --no-cpu-baseline --streams $st 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],3))"; done; done
for m in 0 1 2; do echo "== 1 stream MI_NODE_COLS=$m timeline"; MI_NODE_COLS=$m python scripts/chains_timeline.py --steps 8 --warmup 3 --streams 1 2>&1 | sed -n 3,12p; done
