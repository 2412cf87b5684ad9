cd $GRAFT_REPO_ROOT
hipcc --offload-arch=gfx950 -O3 scripts/force_fwd_repro.hip -o /tmp/ffr 2>/dev/null
{
for kind in 6 8 9; do /tmp/ffr 3000 0 3 $kind 0 0 | head -2; done
} 2>&1 | tee gpurun_out/r4_force_fwd_repro5.log
