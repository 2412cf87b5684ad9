#!/bin/bash
# pair-mode first edge GEMM: the shipped plane GEMM (two accumulator sets, 128 x 128 tiles) vs form E (one set, 128 x 256 tiles, the sine half twice)
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_forward.py -x -q -k "node_chain_launch or north_star or full_size" 2>&1 | tail -4
for m in 0 4 0 4; do for st in 4 1; do MI_EDGE1_FUSED=$m timeout 600 python bench.py --steps 40 --warmup 5 --streams $st --no-cpu-baseline --no-counters 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('edge1_fused=$m chains=$st', round(d['value'],3), 'structures/s', round(d['ms_per_step'],3), 'ms/step', d['config']['final_state_finite'], d['config']['fp16_plane_saturation_events'])"; done; done
