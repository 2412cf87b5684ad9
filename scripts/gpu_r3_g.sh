#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_forward.py -x -q -k "node_chain or north_star or full_size" 2>&1 | tail -15
for cfg in "0 4" "1 4" "0 4" "1 4" "0 1" "1 1"; do
  set -- $cfg
  echo "edge2=$1 streams=$2: $(MI_EDGE2_FUSED=$1 timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --streams $2 2>/dev/null | cut -c75-110)"
done
python scripts/dbg_mg_sat.py 2>&1 | grep -v amdgpu.ids
timeout 1800 python -m pytest tests/test_gpu_mattergen.py -q -s -k "benchmark_size_four or benchmark_size_fine" 2>&1 | grep -E "MEASURED|passed|failed|Error|assert" | head -20
