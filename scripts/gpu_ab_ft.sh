#!/bin/bash
# A/B with the fine-tune and MatterGen-shaped lines too: this tree against variants/libmatinvent_hip_$1.so
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
V=${1:-before}
timeout 1800 python -m pytest tests/test_gpu_forward.py tests/test_gpu_sampler.py tests/test_gpu_train.py tests/test_gpu_gemm.py -x -q -m gpu 2>&1 | tail -2
for rep in 1 2; do for v in $V now; do
  if [ $v = now ]; then unset MI_LIB_PATH; else export MI_LIB_PATH=$GRAFT_REPO_ROOT/matinvent_amd/lib/variants/libmatinvent_hip_$v.so; fi
  echo -n "$v headline: "; python bench.py --steps 20 --warmup 3 --no-counters --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],3))"
  echo -n "$v sample-default: "; python bench.py --mode sample-default --no-counters --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],2))"
  echo -n "$v ft: "; python bench.py --mode ft --no-cpu-baseline --no-counters 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],3))"
  echo -n "$v ft-default: "; python bench.py --mode ft-default --no-cpu-baseline --no-counters 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1))"
  echo -n "$v mg-sample: "; python bench.py --mode mg-sample --no-cpu-baseline --no-counters 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],3), round(d['ms_per_step'],2))"
done; done
