#!/bin/bash
# round 3: the node-chain launch -- parity first, then A/B on the headline line (same box, same session)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_forward.py -x -q -k "node_chain or ragged or north_star or full_size" 2>&1 | tail -15 > gpurun_out/r3_node_pytest.log
cat gpurun_out/r3_node_pytest.log
for nf in 0 1 2 1 0; do
  MI_NODE_FUSED=$nf timeout 300 python bench.py --steps 40 --warmup 5 > gpurun_out/r3_node_b$nf.json 2> gpurun_out/r3_node_b$nf.err
  echo "node_fused=$nf: $(cut -c1-160 gpurun_out/r3_node_b$nf.json)"
done
