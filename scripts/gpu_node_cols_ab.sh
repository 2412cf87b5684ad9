#!/bin/bash
# The column-split node chain (MI_NODE_COLS: 0 = row-block forms of round 4, 1 = one launch per stage, 2 = one launch with hand-overs;
# + 256 (b_max + 1): the workgroup count up to which LayerNorm + projections take one column group per workgroup): parity, then the headline /
# the reference's default sampling batch alternating on one box, then the chains' timeline.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_forward.py -x -q -m gpu -k "node_chain" 2>&1 | tail -3
for rep in 1 2; do for m in 0 1 257 2 258; do
  echo -n "MI_NODE_COLS=$m headline: "; MI_NODE_COLS=$m timeout 300 python bench.py --steps 20 --warmup 3 --no-counters --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],3), round(d['roofline']['frac'],3))"
  echo -n "MI_NODE_COLS=$m sample-default: "; MI_NODE_COLS=$m timeout 300 python bench.py --mode sample-default --steps 20 --warmup 3 --no-counters --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],3))"
done; done
for m in 0 1 257; do echo -n "MI_NODE_COLS=$m 1 stream: "; MI_NODE_COLS=$m timeout 300 python bench.py --steps 20 --warmup 3 --streams 1 --no-counters --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],3))"; done
for m in 1 257; do echo "== MI_NODE_COLS=$m chains timeline"; MI_NODE_COLS=$m python scripts/chains_timeline.py 2>&1 | sed -n 3,7p; done
echo "== 1 stream MI_NODE_COLS=1 timeline"; MI_NODE_COLS=1 python scripts/chains_timeline.py --steps 8 --warmup 3 --streams 1 2>&1 | sed -n 3,6p
