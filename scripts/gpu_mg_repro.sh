#!/bin/bash
# round 4, review item 2: (a) force_fwd_kernel alone on frozen inputs under concurrency; (b) the four-chain forward check at HEAD and with
# HSA_ENABLE_SDMA=0 / GPU_MAX_HW_QUEUES=4
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_mattergen.py -x -q -k "not benchmark_size" 2>&1 | tail -6
hipcc --offload-arch=gfx950 -O3 scripts/force_fwd_repro.hip -o /tmp/ffr 2>/dev/null
for mode in 0 1; do timeout 300 /tmp/ffr 10000 $mode 3; done 2>&1 | tee gpurun_out/r4_force_fwd_repro.log
timeout 600 python scripts/mg_concurrent_forward_check.py 2>&1 | grep -v Warning | tail -32 > gpurun_out/r4_mg_conc_default.log; tail -30 gpurun_out/r4_mg_conc_default.log
HSA_ENABLE_SDMA=0 timeout 600 python scripts/mg_concurrent_forward_check.py 2>&1 | grep -v Warning | tail -30 > gpurun_out/r4_mg_conc_nosdma.log; grep -c identical gpurun_out/r4_mg_conc_nosdma.log; grep "concurrent" gpurun_out/r4_mg_conc_nosdma.log | grep -v identical | head -5
