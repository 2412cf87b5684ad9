#!/bin/bash
# The column-split node chain (MI_NODE_COLS: 0 = row-block forms of round 4, 1 = one launch per stage, 2 = one launch with hand-overs):
# parity, then the headline / the reference's default sampling batch / the fine-tune line alternating on one box.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_forward.py -x -q -m gpu -k "node_chain" 2>&1 | tail -5
for rep in 1 2; do for m in 0 1 2; do
  echo "== MI_NODE_COLS=$m headline"; MI_NODE_COLS=$m timeout 300 python bench.py --steps 20 --warmup 3 --no-counters --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
  echo "== MI_NODE_COLS=$m sample-default"; MI_NODE_COLS=$m timeout 300 python bench.py --mode sample-default --steps 20 --warmup 3 --no-counters --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done
for m in 0 1 2; do echo "== MI_NODE_COLS=$m chains timeline"; MI_NODE_COLS=$m python scripts/chains_timeline.py 2>&1 | sed -n 3,9p; done
for m in 0 1; do echo "== MI_NODE_COLS=$m ft"; MI_NODE_COLS=$m timeout 600 python bench.py --mode ft --no-cpu-baseline --no-counters 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
