#!/bin/bash
# node chain as two launches (phase A, then LayerNorm + the three projection passes on three workgroups per row block) for small / medium batches; alternating on one box
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_gpu_forward.py tests/test_gpu_sampler.py tests/test_gpu_train.py tests/test_gpu_pipeline.py tests/test_gpu_knn.py -x -q 2>&1 | tail -3
for rep in 1 2; do for f in 1 0; do
MI_NODE_SPLIT=$f python bench.py --mode sample-default --steps 60 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('split=$f sample-default', round(d['value'],2), round(d['ms_per_step'],3))"
for st in 4 1; do MI_NODE_SPLIT=$f timeout 600 python bench.py --steps 40 --warmup 5 --streams $st --no-cpu-baseline --no-counters 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('split=$f chains=$st', round(d['value'],3), 'structures/s', round(d['ms_per_step'],3), 'ms/step')"; done
MI_NODE_SPLIT=$f python bench.py --mode ft --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('split=$f ft', round(d['value'],1))"
MI_NODE_SPLIT=$f python bench.py --mode ft-default --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('split=$f ft-default', round(d['value'],1))"
done; done
