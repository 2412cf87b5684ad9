#!/bin/bash
# edge_gemm1b's k-loop by ablation (timing only, wrong results; MI_DBG_E1_LOOP bits: 1 = no MFMAs, 2 = no LDS-DMA of the Fourier operand, 4 = no weight
# ring loads, 8 = no LDS fragment reads): the kernel's duration alone on the chip (one chain of 256 crystals) and its phase clock.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in default l1 l2 l4 l8 l3 l15; do
  if [ $v = default ]; then unset MI_LIB_PATH; else export MI_LIB_PATH=$GRAFT_REPO_ROOT/matinvent_amd/lib/variants/libmatinvent_hip_$v.so; fi
  echo "== $v"; python scripts/chains_timeline.py --steps 6 --warmup 2 --streams 1 2>&1 | grep -E "edge_gemm1b|edge_gemm2b" | cut -c1-130
  python scripts/edge2_phases.py 256 2>&1 | tail -4 | head -3
done
