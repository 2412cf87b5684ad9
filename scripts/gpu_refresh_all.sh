#!/bin/bash
# round-end evidence at one HEAD: whole GPU suite, rocprofv3 summaries (headline + MatterGen-shaped sampler), every bench line
HEAD=${1:-unknown}
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r2_gpu_pytest.log
bash scripts/profile_round.sh r2 $HEAD > gpurun_out/r2_profile_round.log 2>&1
bash scripts/gpu_mg_prof.sh $HEAD > gpurun_out/r2_mg_prof.log 2>&1
for mode in mg-sample sample-default ft-default mg-ft; do
  python bench.py --mode $mode $([ $mode = mg-ft ] && echo --mg-batch 256) 2> gpurun_out/r2_bench_$mode.err | tail -1 > gpurun_out/r2_bench_$mode.json
done
MI_DEBUG_OPTIME=1 python bench.py --mode mg-sample --steps 1 --warmup 1 --no-cpu-baseline 2> gpurun_out/r2_mg_optime.log > /dev/null
cat gpurun_out/r2_gpu_pytest.log
for f in default_steps20 default finetune mg-sample sample-default ft-default mg-ft; do echo "$f: $(cut -c1-160 gpurun_out/r2_bench_$f.json)"; done
