#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_forward.py -x -q -k "node_chain or ragged or north_star" 2>&1 | tail -5
python scripts/node_chain_phases.py 64 2>&1 | tail -13
python scripts/node_chain_phases.py 256 2>&1 | tail -13
for cfg in "0 4" "1 4" "2 4" "1 4" "0 4" "0 1" "1 1" "2 1"; do
  set -- $cfg
  echo "node_fused=$1 streams=$2: $(MI_NODE_FUSED=$1 timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --streams $2 2>/dev/null | cut -c75-110)"
done
for nf in 0 1 0 1; do
  echo "ft node_fused=$nf: $(MI_NODE_FUSED=$nf timeout 300 python bench.py --mode ft --no-cpu-baseline 2>/dev/null | cut -c55-100)"
done
