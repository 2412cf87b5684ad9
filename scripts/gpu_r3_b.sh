#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 scripts/l2_stream.hip -o /tmp/l2_stream 2>/dev/null && /tmp/l2_stream > gpurun_out/r3_l2_stream.log 2>&1
cat gpurun_out/r3_l2_stream.log
for nf in 0 1 2 1 0; do
  echo "sample-default node_fused=$nf: $(MI_NODE_FUSED=$nf timeout 300 python bench.py --mode sample-default --steps 100 2>/dev/null | cut -c1-150)"
done
for nf in 0 1 1 0; do
  echo "ft-default node_fused=$nf: $(MI_NODE_FUSED=$nf timeout 300 python bench.py --mode ft-default --steps 200 2>/dev/null | cut -c1-150)"
done
for nf in 0 1; do
  echo "ft node_fused=$nf: $(MI_NODE_FUSED=$nf timeout 300 python bench.py --mode ft --no-cpu-baseline 2>/dev/null | cut -c1-150)"
done
