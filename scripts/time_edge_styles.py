"""Denoising-step time of the sampler on the benchmark workload (B = 256 x 20 atoms, random-init network) for the fc and knn edge
styles, on one and on four concurrent chains.  Usage (GPU box): PYTHONPATH=. python scripts/time_edge_styles.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import bench
from matinvent_amd.diffcsp import DiffCSPModule
dev = torch.device('cuda')
if os.environ.get("MI_KNN_NOSYNC", "") != "":   # 0: the synchronising graph build of rounds 1-5 inside the chain (A/B of round 6's sync-free build)
    from matinvent_amd import _lib
    _lib.load().mi_debug_set_knn_nosync(int(os.environ["MI_KNN_NOSYNC"]))
def build(style):
    torch.manual_seed(bench.SEED_W)
    m = DiffCSPModule(decoder=dict(hidden_dim=bench.H, num_layers=bench.L, num_freqs=bench.F, ln=True, edge_style=style, max_neighbors=20),
                      beta_scheduler=dict(timesteps=bench.T, scheduler_mode="cosine"),
                      sigma_scheduler=dict(timesteps=bench.T, sigma_begin=0.005, sigma_end=0.5, sigmas_norm=np.load(bench.SIGMAS_NORM)), device=dev)
    with torch.no_grad():
        v = m.decoder.views()
        for k in ("coord_out.weight", "lattice_out.weight", "type_out.weight", "type_out.bias"):
            v[k].mul_(bench.HEAD_SCALE)
    m.decoder.mark_dirty()
    return m
for style in ("fc", "knn"):
    m = build(style)
    for S in (1, 4):
        class Counts:
            num_atoms = torch.tensor([bench.NATOM] * bench.B)
        cb = Counts()
        K = 100
        m.sample(cb, seed=1, t_start=bench.T, t_stop=bench.T - 5, step_lr=bench.STEP_LR, streams=S)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m.sample(cb, seed=2, t_start=bench.T, t_stop=bench.T - K, step_lr=bench.STEP_LR, streams=S)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
        if style == "knn":
            m.check_graph()
        print(f"{style} streams {S}: {dt*1e3:.2f} ms per denoising step -> {bench.B/(dt*bench.T):.1f} structures/s")
