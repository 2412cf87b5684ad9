cd $GRAFT_REPO_ROOT
hipcc --offload-arch=gfx950 -O3 scripts/force_fwd_repro.hip -o /tmp/ffr 2>/dev/null
hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize scripts/force_fwd_repro.hip -o /tmp/ffr_noslp 2>/dev/null
{
echo "== default build (the head uses v_pk_*_f32)"; /tmp/ffr 3000 0 3 0 0 0 | head -3
echo "== -fno-slp-vectorize build (no packed fp32 op in the head)"; /tmp/ffr_noslp 3000 0 3 0 0 0 | head -3
/tmp/ffr_noslp 3000 0 1 0 0 0 | head -2
echo "== default build, busy kinds 3 (8 accumulators, no LDS) and 4 (4 accumulators, LDS)"
/tmp/ffr 3000 0 3 3 0 0 | head -3
/tmp/ffr 3000 0 3 4 0 0 | head -3
} 2>&1 | tee gpurun_out/r4_force_fwd_repro3.log
