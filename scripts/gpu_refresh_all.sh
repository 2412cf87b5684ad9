#!/bin/bash
# round-end evidence at one HEAD (on the GPU box): whole GPU suite + the driver-shaped line, rocprofv3 summaries (headline, fine-tune,
# MatterGen-shaped sampler), every secondary bench line.  usage: bash scripts/gpu_refresh_all.sh <tag, e.g. r3> <git head>
TAG=${1:-r3}; HEAD=${2:-unknown}
cd $GRAFT_REPO_ROOT
bash scripts/gpu_full.sh > gpurun_out/${TAG}_gpu_full.log 2>&1
bash scripts/profile_round.sh $TAG $HEAD > gpurun_out/${TAG}_profile_round.log 2>&1
bash scripts/gpu_ft_prof.sh > gpurun_out/${TAG}_ft_prof.log 2>&1
bash scripts/gpu_mg_prof.sh $HEAD $TAG > gpurun_out/${TAG}_mg_prof.log 2>&1
bash scripts/gpu_secondary.sh > gpurun_out/${TAG}_secondary.log 2>&1
tail -12 gpurun_out/${TAG}_gpu_full.log; cat gpurun_out/${TAG}_secondary.log
