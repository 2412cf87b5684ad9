#!/bin/bash
# Marginal cost of the pair-mode Fourier operand kernel inside the four-chain step (timing ablation, results garbage): MI_SKIP=16 against 0, alternating.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { timeout 300 python bench.py --steps 100 --warmup 5 --no-counters --no-cpu-baseline $EXTRA 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],3))"; }
for i in 1 2 3; do for sk in 0 16; do echo -n "MI_SKIP=$sk: "; MI_SKIP=$sk run; done; done | tee gpurun_out/r5_skip_fourier.log
