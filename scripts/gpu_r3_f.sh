#!/bin/bash
cd $GRAFT_REPO_ROOT
python scripts/dbg_mg_sat.py 2>&1 | grep -v amdgpu.ids
timeout 1800 python -m pytest tests/test_gpu_mattergen.py -q -s -k "benchmark_size_four or benchmark_size_fine" 2>&1 | grep -E "MEASURED|passed|failed|Error|assert" | head -20
