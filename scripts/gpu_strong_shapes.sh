#!/bin/bash
# Strong-scaling evidence from ONE GPU (round 6, review item 3): the per-rank shapes of a global batch of 256 sharded over 1 / 2 / 4 / 8 ranks
# (B = 256 / 128 / 64 / 32 crystals x 20 atoms), sampler and fine-tune, each at the sampler's automatic chain count and at 1 / 2 / 4 chains.
# Writes gpurun_out/r6_strong_shapes.jsonl (one bench line per run, tagged); scripts/strong_shapes_summary.py turns it into profiles/r6_strong_shapes.json.
# usage (through gpurun): bash scripts/gpu_strong_shapes.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_strong_shapes.jsonl
: > $O
for B in 256 128 64 32; do
  if [ $B = 256 ]; then SS="0"; GG="0"; else SS="0 1 2 4"; GG="0 1 2 3"; fi
  for S in $SS; do
    python3 bench.py --batch $B --streams $S --steps 100 --warmup 5 --no-cpu-baseline --no-counters 2>> gpurun_out/r6_strong_shapes.err | sed "s/^{/{\"tag\": \"sample B=$B streams=$S\", /" >> $O
  done
  for G in $GG; do
    if [ $G = 0 ]; then GA=""; else GA="--ft-groups $G"; fi
    python3 bench.py --mode ft --batch $B $GA --steps 50 --warmup 3 --no-cpu-baseline --no-counters 2>> gpurun_out/r6_strong_shapes.err | sed "s/^{/{\"tag\": \"ft B=$B groups=$G\", /" >> $O
  done
done
python3 scripts/strong_shapes_summary.py $O gpurun_out/r6_strong_shapes.json
