#!/bin/bash
# the reference's default sampling batch (192 ragged crystals): the first edge GEMM's forms at its small launches
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for f in 9 1 0; do MI_EDGE1_FUSED=$f python bench.py --mode sample-default --steps 60 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('edge1=$f sample-default', round(d['value'],2), round(d['ms_per_step'],3))"; done; done
